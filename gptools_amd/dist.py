"""One-process-per-GPU block-cyclic panel Cholesky for the log-marginal-likelihood at large N.

The reference has no distributed code at all (its only parallelism is a process
pool over independent hyperparameter samples); what must match is the *result* of
``GaussianProcess.compute_K_L_alpha_ll`` (ref: gptools/gaussian_process.py:1418-1469): ``ll`` and
``sum(log diag L)``.  Design:

  * 1-D block-cyclic distribution of block columns: block column J (``nb`` wide) lives on rank
    ``J % world`` as columns ``[lj*nb, (lj+1)*nb)`` of that rank's local (NP x nloc*nb) row-major
    matrix, full height.  Every rank holds X, n, y (a few hundred KB) and builds its own block
    columns of K_tot with the fused K-builder: K never crosses xGMI.
  * Right-looking factorisation: the owner factors panel k (diagonal block + TRSM of the rows below,
    ``gpt_dev_potrf_panel``) in a contiguous (m x nb) buffer and broadcasts it (RCCL over xGMI via
    ``torch.distributed.broadcast``); every rank applies it to its local trailing block columns
    with the fp64-MFMA SYRK/GEMM (``gpt_dev_gemm_nt``).  Look-ahead: the owner of panel k+1 updates
    and factors that block column first and starts its broadcast asynchronously, before anyone
    finishes applying panel k, so the transfer and the latency-bound panel hide behind the updates.
  * ``z = L^-1 y`` rides along as the augmented row N of the matrix, so the only
    other collective is one all-reduce of three scalars (log-det part, z.z part, info).
  * Two ways to move a panel (``exchange=``): one broadcast, or scatter + all-gather (every xGMI link of the root carries
    1/world of the panel, then all links carry the all-gather) for panels above ``sag_min_bytes``.
  * :class:`GridLML` (below) is the same evaluation on a ``P_r x P_c`` process grid (2-D block-cyclic): the per-panel
    serial work is split over a process column.  (A row-chunked "pipelined" 1-D schedule existed in rounds 1-3; modelled at
    22 % of the 8-GPU peak it was removed: scratch/attic/dist_pipelined_schedule.py.txt.)

All dense work goes through a small ``ops`` object.  The product implementation is
:class:`HipPanelOps` (C ABI of libgpt_hip.so on CUDA tensors; raises without a GPU).  The
multi-process CPU tests inject a numpy implementation from ``tests/`` to exercise the
partitioning / communication logic under the ``gloo`` backend.
"""
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

__all__ = ["PanelOps", "HipPanelOps", "DistributedLML"]

BIG_PIVOT = 1e300


class _NoEvent(object):
    """Stand-in for a stream event where everything runs in program order (CPU test ops)."""

    def record(self):
        pass

    def wait(self):
        pass


class PanelOps(object):
    """Interface of the dense local operations.  Two in-order queues exist per rank: ``"panel"`` (the
    latency-bound chain: staging, look-ahead column update, panel factorisation, broadcasts) and ``"main"``
    (K build and trailing updates).  This base class is the serial version: one queue, events are no-ops."""
    device = torch.device("cpu")

    def queue(self, q):
        import contextlib
        return contextlib.nullcontext()

    def new_event(self):
        return _NoEvent()

    def synchronize(self):
        pass

    def new_timing_event(self):
        """An event whose position on the device timeline can be read back (``elapsed_ms``); None where there is no
        device timeline (CPU test ops)."""
        return None

    # -- small data movers / reductions.  The defaults are plain torch (the CPU test ops inherit them); the product
    #    ops override all three with library kernels so that the queues of a GPU rank carry no framework kernels.
    def copy2d(self, dst, src, q="panel"):
        """dst <- src, two 2-D views of equal shape (staging a block column into a panel buffer)."""
        dst.copy_(src)

    def pad_block(self, A, lj, c0, nb, N, NP, y, big, row_shift=0):
        """Rows [N, NP) of the local block column ``lj`` (global first column ``c0``): augmented row y^T, unit diagonal on
        the padding, ``big`` under the augmented row, zeros elsewhere.  ``row_shift``: global row r of
        the matrix is local row ``r + row_shift`` of ``A`` (the 2-D layout holds only some block rows)."""
        blk = A[N + row_shift:NP + row_shift, lj * nb:(lj + 1) * nb]
        blk.zero_()
        c1 = min(c0 + nb, N)
        if c0 < N:
            blk[0, :c1 - c0] = y[c0:c1]
        p0 = max(c0, N)
        if p0 < c0 + nb:
            idx = torch.arange(p0, c0 + nb, device=A.device)
            A[idx + row_shift, lj * nb + (idx - c0)] = 1.0
            if c0 <= N < c0 + nb:
                A[N + row_shift, lj * nb + (N - c0)] = big

    def panel_scalars(self, buf, w, zrow, red, q="panel"):
        """red[0] += sum(log diag(buf[:w, :w])); if zrow >= 0 also red[1] += |buf[zrow, :w]|^2."""
        red[0] += torch.log(torch.diagonal(buf[:w, :w])).sum()
        if zrow >= 0:
            z = buf[zrow, :w]
            red[1] += (z * z).sum()

    def row_sumsq(self, row, red, q="panel"):
        """red[1] += |row|^2 (``row`` a contiguous 1-D view: a piece of the augmented row)."""
        red[1] += (row * row).sum()

    def gemm_nt_gridstair(self, m, nseg, seg_cols, k, alpha, A, lda, B, ldb, off, num, den, base, beta, C, ldc, q="main"):
        """Default of the 2-D block-cyclic trailing update (gpt_dev_gemm_nt_gridstair): one ``gemm_nt`` per column segment, a
        lower trapezoid where the segment starts on a diagonal block of the matrix, a rectangle otherwise."""
        for s in range(nseg):
            v = off + s * num
            r = (-((-v) // den) - base) * seg_cols
            if r >= m:
                continue
            self.gemm_nt(m - r, seg_cols, k, alpha, A + r * lda * 8, lda, B + s * seg_cols * ldb * 8, ldb, beta,
                         C + (r * ldc + s * seg_cols) * 8, ldc, 1 if v % den == 0 else 0, q=q)

    def gemm_nt_stair(self, m, nseg, seg_cols, k, alpha, A, lda, B, ldb, b_stride, row_step, beta, C, ldc, q="main"):
        """Default: one lower-trapezoid ``gemm_nt`` per column segment (8-byte elements)."""
        for s in range(nseg):
            r = s * row_step
            self.gemm_nt(m - r, seg_cols, k, alpha, A + r * lda * 8, lda, B + s * b_stride * ldb * 8, ldb, beta,
                         C + (r * ldc + s * seg_cols) * 8, ldc, 1, q=q)


class _StreamEvent(object):
    """An event on the queue that is current in ``ops`` (HipPanelOps.queue keeps the stream object at hand: looking it up through
    ``torch.cuda.current_stream()`` for every record / wait was a third of the step loop's host time, round 5)."""
    __slots__ = ("ev", "ops")

    def __init__(self, ops, timing=False):
        self.ev = torch.cuda.Event(enable_timing=timing)
        self.ops = ops

    def elapsed_ms(self, later):
        return self.ev.elapsed_time(later.ev)

    def record(self):
        self.ev.record(self.ops._cur or torch.cuda.current_stream())

    def wait(self):
        (self.ops._cur or torch.cuda.current_stream()).wait_event(self.ev)


class _QueueCtx(object):
    """``with ops.queue(q)``: torch's current stream AND the ops' own note of it (nested uses restore the outer queue)."""
    __slots__ = ("ops", "st", "inner", "prev")

    def __init__(self, ops, st):
        self.ops, self.st = ops, st
        self.inner = torch.cuda.stream(st)

    def __enter__(self):
        self.prev = self.ops._cur
        self.ops._cur = self.st
        self.inner.__enter__()
        return self

    def __exit__(self, *exc):
        self.inner.__exit__(*exc)
        self.ops._cur = self.prev
        return False


class HipPanelOps(PanelOps):
    """Dense local operations on CUDA tensors through the device API of include/gpt_hip.h.

    Two library contexts, one per queue: the main one owns the library's CU-masked stream (the trailing updates leave
    a few CUs to the panel kernels), the panel one runs on a high-priority torch stream.
    torch's tensor ops and the RCCL collectives are issued with the matching stream current, so each queue is one
    ordered HIP stream; cross-queue edges are torch events."""

    def __init__(self, device):
        if not torch.cuda.is_available():
            raise _lib.GPTBackendError("HipPanelOps needs a GPU (gptools_amd has no CPU fallback)")
        self.device = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        torch.cuda.set_device(self.device)
        self.lib = _lib.load()
        self.ctx_main = _lib.Context(self.device.index)
        self.main_stream = torch.cuda.ExternalStream(int(self.ctx_main.stream), device=self.device)
        self.panel_stream = torch.cuda.Stream(self.device, priority=-1)
        self.ctx_panel = _lib.Context(self.device.index, stream=self.panel_stream.cuda_stream)
        # "recv": where a rank that does not own a panel posts its side of the exchange -- an otherwise empty stream,
        # so that the receive never queues up behind this rank's own panel work
        self.recv_stream = torch.cuda.Stream(self.device, priority=-1)
        # (a context of its own for "recv": GridLML runs the bulk half of its panel work there -- GEMMs, copies, reductions)
        self.ctx_recv = _lib.Context(self.device.index, stream=self.recv_stream.cuda_stream)
        self._cur = None                       # the stream of the innermost ``with self.queue(q)`` (see _StreamEvent)
        self._ctx = {"main": self.ctx_main, "panel": self.ctx_panel, "recv": self.ctx_recv}
        self._stream = {"main": self.main_stream, "panel": self.panel_stream, "recv": self.recv_stream}
        for c in self._ctx.values():
            c.set_option("lookahead", 0)
        # Residency cap of the main queue's trailing updates (context option dev_gemm_pad, GPT_DIST_MAIN_PAD bytes of dummy LDS per
        # workgroup; 24576 = two workgroups of the update per CU instead of five).  Isolated -- one trapezoid update, the chain of one
        # panel looping beside it -- the cap is free for the update and takes the chain from 1.95 x to 1.43 x its stand-alone time
        # (profiles/r06_chain_contention.txt); IN the engines it loses: the staircase updates run 10 % slower with
        # it (C4 at world size 1: 218 -> 242 ms) and the converged 8-rank replay model goes from 44.2 to 45.1 ms.  Off by default.
        self.ctx_main.set_option("dev_gemm_pad", int(os.environ.get("GPT_DIST_MAIN_PAD", "0")))
        # everything the panel context launches sits on the chain and shares CUs with the main context's trailing update:
        # its GEMM main loops keep a raised wave priority (gemm.hip; the TRSM / fused kernels carry theirs themselves)
        self.ctx_panel.set_option("gemm_prio", int(os.environ.get("GPT_DIST_PANEL_PRIO", "2")))
        self.ctx_recv.set_option("gemm_prio", int(os.environ.get("GPT_DIST_PANEL_PRIO", "2")))

    def queue(self, q):
        return _QueueCtx(self, self._stream[q])

    def new_event(self):
        return _StreamEvent(self)

    def new_timing_event(self):
        return _StreamEvent(self, timing=True)

    def synchronize(self):
        for st in self._stream.values():
            st.synchronize()

    def kbuild_block(self, kernel_id, params, X, n, r0, r1, c0, c1, err_y, noise_var, diag_add, out, ld):
        """out[(i - r0) * ld + (j - c0)] = K_tot[i][j] for i in [r0, r1), j in [c0, c1) (global indices)."""
        params = _lib.f64(params)
        D = X.shape[1]
        esz_d, esz_i = 8, 4
        _lib.check(self.lib.gpt_dev_kbuild(
            self.ctx_main.handle, int(kernel_id), _lib.dptr(params), len(params),
            X.data_ptr() + r0 * D * esz_d, n.data_ptr() + r0 * D * esz_i, r1 - r0,
            X.data_ptr() + c0 * D * esz_d, n.data_ptr() + c0 * D * esz_i, c1 - c0, D,
            -1, 1, None, 1, r0, c0, err_y.data_ptr(), float(noise_var), float(diag_add), out, ld))

    def potrf_panel(self, m, nb, A, lda, invd, info, info_base):
        _lib.check(self.lib.gpt_dev_potrf_panel(self.ctx_panel.handle, m, nb, A, lda, invd.data_ptr(),
                                                info.data_ptr(), info_base))

    def trsm_rlt(self, m, nb, L, ldl, invd, B, ldb, q="panel"):
        """B (m x nb) <- B L^-T on queue ``q``, L the factored diagonal block of the same panel."""
        _lib.check(self.lib.gpt_dev_trsm_rlt(self._ctx[q].handle, m, nb, L, ldl, invd.data_ptr(), B, ldb))

    def trinv(self, nb, L, ldl, invd, W, ldw, q="panel"):
        """W (nb x nb) <- L^-1 on queue ``q`` (gpt_dev_trinv): the TRSM of a tall chunk then is one GEMM."""
        _lib.check(self.lib.gpt_dev_trinv(self._ctx[q].handle, nb, L, ldl, invd.data_ptr(), W, ldw))

    def gemm_nt(self, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, q="main"):
        _lib.check(self.lib.gpt_dev_gemm_nt(self._ctx[q].handle, m, n, k, float(alpha), A, lda, B, ldb, float(beta),
                                            C, ldc, int(tri)))

    def copy2d(self, dst, src, q="panel"):
        assert dst.shape == src.shape and dst.stride(1) == 1 and src.stride(1) == 1
        # (every queue has a library context of its own, "recv" included)
        _lib.check(self.lib.gpt_dev_copy2d(self._ctx[q].handle, src.shape[0], src.shape[1], src.data_ptr(), src.stride(0),
                                           dst.data_ptr(), dst.stride(0)))

    def pad_block(self, A, lj, c0, nb, N, NP, y, big, row_shift=0):
        # (row_shift: the library addresses row r of the block column as base + r * lda; a shifted base makes that the local row)
        _lib.check(self.lib.gpt_dev_pad_block(self.ctx_main.handle, _ptr(A, 0, lj * nb) + row_shift * A.stride(0) * 8, A.stride(0),
                                              c0, nb, N, NP, y.data_ptr(), float(big)))

    def kbuild_rect(self, kernel_id, params, Xi, ni, r0, r1, Xj, nj, c0, c1, out, ld):
        """out[(i - r0) * ld + (j - c0)] = k(Xi[i], Xj[j]) for i in [r0, r1), j in [c0, c1): a plain rectangle, no diagonal
        terms (the 2-D layout's blocks below the diagonal; rows of ``Xi`` are the rank's own block rows, gathered)."""
        params = _lib.f64(params)
        D = Xi.shape[1]
        _lib.check(self.lib.gpt_dev_kbuild(
            self.ctx_main.handle, int(kernel_id), _lib.dptr(params), len(params),
            Xi.data_ptr() + r0 * D * 8, ni.data_ptr() + r0 * D * 4, r1 - r0,
            Xj.data_ptr() + c0 * D * 8, nj.data_ptr() + c0 * D * 4, c1 - c0, D,
            -1, 1, None, 0, 0, 0, None, 0.0, 0.0, out, ld))

    def row_sumsq(self, row, red, q="panel"):
        _lib.check(self.lib.gpt_dev_row_sumsq(self._ctx[q].handle, row.data_ptr(), row.numel(), red.data_ptr() + 8))

    def gemm_nt_gridstair(self, m, nseg, seg_cols, k, alpha, A, lda, B, ldb, off, num, den, base, beta, C, ldc, q="main"):
        _lib.check(self.lib.gpt_dev_gemm_nt_gridstair(self._ctx[q].handle, m, nseg, seg_cols, k, float(alpha), A, lda, B, ldb,
                                                      off, num, den, base, float(beta), C, ldc))

    def panel_scalars(self, buf, w, zrow, red, q="panel"):
        _lib.check(self.lib.gpt_dev_panel_scalars(self._ctx[q].handle, buf.data_ptr(), buf.stride(0), w, zrow,
                                                  red.data_ptr()))

    def gemm_nt_stair(self, m, nseg, seg_cols, k, alpha, A, lda, B, ldb, b_stride, row_step, beta, C, ldc, q="main"):
        """All block columns a rank owns right of the panel in ONE launch (gpt_dev_gemm_nt_stair)."""
        _lib.check(self.lib.gpt_dev_gemm_nt_stair(self._ctx[q].handle, m, nseg, seg_cols, k, float(alpha), A, lda, B, ldb,
                                                  b_stride, row_step, float(beta), C, ldc))


def _ptr(t, row, col):
    """Address of element (row, col) of a 2-D row-major tensor."""
    return t.data_ptr() + (row * t.stride(0) + col) * t.element_size()


# ======================================================================================================
# Compiled schedules (round 6): the step loop recorded once as an op list, replayed per evaluation
# ======================================================================================================
# What a rank does per evaluation -- which library kernels on which buffers, on which queue, in which order, which events and
# which panel exchanges between them -- depends on (N, nb, world, rank, options) only.  ``PlanRecorder`` is a PanelOps that
# executes nothing: driven ONCE through the engine's own step loop it yields the op list; ``CompiledPlan`` replays it
#   * natively: one call of ``gpt_plan_run`` (csrc/api_plan.inc) per evaluation -- a C loop over fixed-size integer records, RCCL
#     called directly from the library on the plan's own communication stream (product path, HipPanelOps), or
#   * through a Python interpreter of the SAME list against any PanelOps + torch.distributed (the gloo / numpy tests, and the
#     ranks-sharing-one-GPU tests, where RCCL cannot run).
OP_RECORD, OP_WAIT, OP_KBUILD, OP_PAD, OP_COPY2D, OP_POTRF_PANEL, OP_TRINV, OP_GEMM, OP_STAIR, OP_SCALARS, OP_BCAST, OP_SCATTER, \
    OP_ALLGATHER, OP_KRECT, OP_ROWSUMSQ, OP_GRIDSTAIR = range(16)
PLAN_W = 20
PLAN_CHANNELS = 8
# queues: the three contexts, then the communication channels ("comm" = channel 0: the 1-D engine's only one, the grid's whole-grid one)
QUEUE_ID = dict({"main": 0, "panel": 1, "recv": 2, "comm": 3}, **{"comm%d" % c: 3 + c for c in range(1, PLAN_CHANNELS)})


def _f64_bits(x):
    import struct
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


class _RecEvent(object):
    """An event of a recorded schedule.  The recorder keeps a vector clock per context queue (in-order streams in every
    executor) and the clock an event was recorded at: a wait whose event is already behind the waiting queue -- recorded on that
    queue itself, waited for before, or ordered through an event the queue waited for since -- is NOT emitted (a third of the
    2-D engine's waits; each one is a hipStreamWaitEvent per evaluation).  An exchange's "arrived" event is a clock component of its
    own: the channel queues are not assumed to complete in order (gloo's asynchronous operations do not)."""
    __slots__ = ("rec", "idx", "vec")

    def __init__(self, rec):
        self.rec, self.idx, self.vec = rec, rec.nevents, None
        rec.nevents += 1

    def record(self):
        rec, q = self.rec, self.rec.cur
        clk = rec.clock.setdefault(q, {})
        clk[q] = clk.get(q, 0) + 1
        self.vec = dict(clk)
        rec.emit(OP_RECORD, q, [self.idx])

    def arrived_after(self, ready):
        """(recorded by a channel queue behind ``ready``)"""
        self.vec = dict(ready.vec)
        self.vec[("arrived", self.idx)] = 1

    def wait(self):
        rec, q = self.rec, self.rec.cur
        if self.vec is None:
            raise RuntimeError("a recorded schedule waits for an event before recording it")
        clk = rec.clock.setdefault(q, {})
        if all(clk.get(k, 0) >= v for k, v in self.vec.items()):
            rec.pruned += 1
            return
        for k, v in self.vec.items():
            if clk.get(k, 0) < v:
                clk[k] = v
        rec.emit(OP_WAIT, q, [self.idx])


class _RecQueue(object):
    __slots__ = ("rec", "q", "prev")

    def __init__(self, rec, q):
        self.rec, self.q = rec, q

    def __enter__(self):
        self.prev, self.rec.cur = self.rec.cur, self.q
        return self

    def __exit__(self, *exc):
        self.rec.cur = self.prev
        return False


class PlanRecorder(PanelOps):
    """A PanelOps that records instead of executing.  ``ops``: list of ``(opcode, queue name, ints, py)`` -- ``ints`` is the native
    record (addresses, sizes, bit patterns of doubles: include/gpt_hip.h "compiled schedules"), ``py`` what the Python interpreter
    hands to the real ops object."""

    def __init__(self, device):
        self.device = device
        self.ops, self.nevents, self.cur = [], 0, "main"
        self.clock, self.pruned = {}, 0         # vector clock per context queue; waits not emitted (see _RecEvent)

    def emit(self, opcode, q, ints, py=None):
        assert len(ints) <= PLAN_W - 2
        self.ops.append((opcode, q, [int(v) for v in ints], py))

    def queue(self, q):
        return _RecQueue(self, q)

    def new_event(self):
        return _RecEvent(self)

    def new_timing_event(self):
        return None

    def synchronize(self):
        pass

    # -- dense ops (the per-evaluation inputs of the K builder -- kernel id, hyperparameters, noise variance, loading -- are NOT
    #    recorded: they are arguments of every replay)
    def kbuild_block(self, kernel_id, params, X, n, r0, r1, c0, c1, err_y, noise_var, diag_add, out, ld):
        self.emit(OP_KBUILD, "main", [r0, r1, c0, c1, out, ld], (r0, r1, c0, c1, out, ld))

    def pad_block(self, A, lj, c0, nb, N, NP, y, big, row_shift=0):
        self.emit(OP_PAD, "main", [_ptr(A, 0, lj * nb) + row_shift * A.stride(0) * 8, A.stride(0), c0, nb, N, NP, y.data_ptr(),
                                   _f64_bits(big)], (A, lj, c0, nb, N, NP, y, big, row_shift))

    def copy2d(self, dst, src, q="panel"):
        assert dst.shape == src.shape and dst.stride(1) == 1 and src.stride(1) == 1
        self.emit(OP_COPY2D, q, [src.shape[0], src.shape[1], src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0)], (dst, src))

    def potrf_panel(self, m, nb, A, lda, invd, info, info_base):
        self.emit(OP_POTRF_PANEL, "panel", [m, nb, A, lda, invd.data_ptr(), info.data_ptr(), info_base],
                  (m, nb, A, lda, invd, info, info_base))

    def trinv(self, nb, L, ldl, invd, W, ldw, q="panel"):
        self.emit(OP_TRINV, q, [nb, L, ldl, invd.data_ptr(), W, ldw], (nb, L, ldl, invd, W, ldw))

    def gemm_nt(self, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, q="main"):
        self.emit(OP_GEMM, q, [m, n, k, _f64_bits(alpha), A, lda, B, ldb, _f64_bits(beta), C, ldc, tri],
                  (m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri))

    def gemm_nt_stair(self, m, nseg, seg_cols, k, alpha, A, lda, B, ldb, b_stride, row_step, beta, C, ldc, q="main"):
        self.emit(OP_STAIR, q, [m, nseg, seg_cols, k, _f64_bits(alpha), A, lda, B, ldb, b_stride, row_step, _f64_bits(beta), C, ldc],
                  (m, nseg, seg_cols, k, alpha, A, lda, B, ldb, b_stride, row_step, beta, C, ldc))

    def panel_scalars(self, buf, w, zrow, red, q="panel"):
        self.emit(OP_SCALARS, q, [buf.data_ptr(), buf.stride(0), w, zrow, red.data_ptr()], (buf, w, zrow, red))

    def kbuild_rect(self, kernel_id, params, Xi, ni, r0, r1, Xj, nj, c0, c1, out, ld):
        self.emit(OP_KRECT, "main", [Xi.data_ptr(), ni.data_ptr(), r0, r1, c0, c1, out, ld], (Xi, ni, r0, r1, Xj, nj, c0, c1, out, ld))

    def row_sumsq(self, row, red, q="panel"):
        self.emit(OP_ROWSUMSQ, q, [row.data_ptr(), row.numel(), red.data_ptr() + 8], (row, red))

    def gemm_nt_gridstair(self, m, nseg, seg_cols, k, alpha, A, lda, B, ldb, off, num, den, base, beta, C, ldc, q="main"):
        self.emit(OP_GRIDSTAIR, q, [m, nseg, seg_cols, k, _f64_bits(alpha), A, lda, B, ldb, off, num, den, base, _f64_bits(beta), C, ldc],
                  (m, nseg, seg_cols, k, alpha, A, lda, B, ldb, off, num, den, base, beta, C, ldc))

    # -- a broadcast on communication channel ``channel`` (a queue name, "comm" / "comm1" ...), root = rank inside that channel's
    #    communicator; ``what`` is handed to the engine's ``_plan_collective`` by the Python interpreter
    def bcast(self, buf, root, channel, what):
        ready, arrived = _RecEvent(self), _RecEvent(self)
        issuing = self.cur
        ready.record()
        self.emit(OP_WAIT, channel, [ready.idx])
        self.emit(OP_BCAST, channel, [buf.data_ptr(), buf.numel(), root], (buf, what, issuing, arrived.idx))
        self.emit(OP_RECORD, channel, [arrived.idx])
        arrived.arrived_after(ready)
        return [arrived]

    # -- the panel exchange: [issuing queue records "ready"] [comm stream waits for it] [collective(s)] [comm stream records
    #    "arrived"]; the returned handle's wait() is a wait for "arrived" on whatever queue is current then
    def exchange(self, buf, src, world, grank, scatter_gather):
        ready, arrived = _RecEvent(self), _RecEvent(self)
        issuing = self.cur
        ready.record()
        self.emit(OP_WAIT, "comm", [ready.idx])
        rows = buf.shape[0]
        if scatter_gather:
            cnt = (rows // world) * buf.shape[1]
            self.emit(OP_SCATTER, "comm", [buf.data_ptr(), cnt, src], (buf, src, issuing, arrived.idx))
            self.emit(OP_ALLGATHER, "comm", [buf.data_ptr(), cnt], None)
        else:
            self.emit(OP_BCAST, "comm", [buf.data_ptr(), buf.numel(), src], (buf, src, issuing, arrived.idx))
        self.emit(OP_RECORD, "comm", [arrived.idx])
        arrived.arrived_after(ready)
        return [arrived]


class CompiledPlan(object):
    """The op list of one rank's evaluation and its two executors (see above)."""

    def __init__(self, recorder):
        # (records of events nobody waits for -- their waits were all pruned, or the step loop keeps them for a later step that
        # never comes -- are dropped: one hipEventRecord less per evaluation each)
        waited = {o[2][0] for o in recorder.ops if o[0] == OP_WAIT}
        self.ops = [o for o in recorder.ops if not (o[0] == OP_RECORD and o[2][0] not in waited)]
        self.nevents = recorder.nevents
        self.handle = None
        self._keep = None

    # ---- native ----
    def build_native(self, hip_ops, X, n, err, channels):
        """``channels``: [(channel index, ranks in it, this rank's position, the torch process group that carries the id)] -- every
        rank lists its channels in the same order (each ``gpt_plan_set_channel`` is collective over that channel's ranks)."""
        lib = hip_ops.lib
        arr = np.zeros((len(self.ops), PLAN_W), dtype=np.int64)
        for i, (opcode, q, ints, _) in enumerate(self.ops):
            arr[i, 0], arr[i, 1] = opcode, QUEUE_ID[q]
            arr[i, 2:2 + len(ints)] = ints
        import ctypes as C
        ctxs = (C.c_void_p * 3)(hip_ops.ctx_main.handle, hip_ops.ctx_panel.handle, hip_ops.ctx_recv.handle)
        out = C.c_void_p()
        _lib.check(lib.gpt_plan_create(3, ctxs, arr.ctypes.data_as(C.POINTER(C.c_int64)), len(self.ops), self.nevents, X.data_ptr(),
                                       n.data_ptr(), int(X.shape[1]), err.data_ptr(), C.byref(out)))
        self.handle, self.lib = out, lib
        self._keep = (X, n, err)
        for ch, nranks, rank, group in channels:
            # the communicator's id: made on the channel's rank 0, handed round through the job's own process group
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_char * 128)()
                _lib.check(lib.gpt_plan_unique_id(buf))
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            if nranks > 1:
                dev_uid = uid.to(hip_ops.device) if dist.get_backend(group) == "nccl" else uid
                dist.broadcast(dev_uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                uid = dev_uid.cpu()
            raw = bytes(uid.numpy().tobytes())
            _lib.check(lib.gpt_plan_set_channel(self.handle, int(ch), int(nranks), int(rank), C.c_char_p(raw)))
        return self

    def run_native(self, kernel_id, params, noise_var, diag_add):
        params = _lib.f64(params)
        _lib.check(self.lib.gpt_plan_run(self.handle, int(kernel_id), _lib.dptr(params), len(params), float(noise_var), float(diag_add)))
        return self.lib.gpt_plan_last_enqueue_ms(self.handle)

    def close(self):
        if self.handle is not None:
            self.lib.gpt_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Python interpreter of the same list ----
    def run_python(self, ops, engine, kernel_id, params, noise_var, diag_add):
        events = [None] * self.nevents
        works = {}
        state = [None, None]                 # the queue that is current, its context manager

        def switch(q):
            # (entering a queue = making its stream current: done when the queue CHANGES, not per op -- the list keeps runs of
            # operations on one queue together)
            if q != state[0]:
                if state[1] is not None:
                    state[1].__exit__(None, None, None)
                state[0], state[1] = q, ops.queue(q)
                state[1].__enter__()

        try:
            for opcode, q, ints, py in self.ops:
                if QUEUE_ID[q] >= 3:
                    # torch.distributed orders a collective behind the stream that is current when it is issued and hands back work
                    # objects: "ready" / the channel stream's own waits have no counterpart, "arrived" is the work objects
                    if opcode in (OP_BCAST, OP_SCATTER):
                        buf, what, issuing, arrived = py
                        switch(issuing)
                        works[arrived] = engine._plan_collective(opcode, buf, what)
                    continue
                switch(q)
                if opcode == OP_WAIT:
                    if ints[0] in works:
                        # (an arrival may be waited for from several queues: the handles stay)
                        for w in works[ints[0]]:
                            if w is not None:
                                w.wait()
                    elif events[ints[0]] is not None:
                        events[ints[0]].wait()
                    # (else: "ready" of an exchange the engine's _plan_collective did not issue, or an "arrived" nobody recorded)
                elif opcode == OP_RECORD:
                    e = events[ints[0]] = ops.new_event()
                    e.record()
                elif opcode == OP_GEMM:
                    ops.gemm_nt(*py, q=q)
                elif opcode == OP_COPY2D:
                    ops.copy2d(py[0], py[1], q=q)
                elif opcode == OP_GRIDSTAIR:
                    ops.gemm_nt_gridstair(*py, q=q)
                elif opcode == OP_STAIR:
                    ops.gemm_nt_stair(*py, q=q)
                elif opcode == OP_KBUILD:
                    r0, r1, c0, c1, out, ld = py
                    ops.kbuild_block(kernel_id, params, engine.X, engine.n, r0, r1, c0, c1, engine.err, noise_var, diag_add, out, ld)
                elif opcode == OP_KRECT:
                    ops.kbuild_rect(kernel_id, params, *py)
                elif opcode == OP_PAD:
                    ops.pad_block(*py)
                elif opcode == OP_POTRF_PANEL:
                    ops.potrf_panel(*py)
                elif opcode == OP_TRINV:
                    ops.trinv(*py, q=q)
                elif opcode == OP_SCALARS:
                    ops.panel_scalars(*py, q=q)
                elif opcode == OP_ROWSUMSQ:
                    ops.row_sumsq(*py, q=q)
                else:
                    raise ValueError("unknown opcode %r" % (opcode,))
        finally:
            if state[1] is not None:
                state[1].__exit__(None, None, None)


class _Arrival(object):
    """One row chunk of a panel becoming readable on this rank: the exchange's work handle (None when no collective was
    issued) and, on the owner, the event after the kernels that produced it.  ``wait()`` orders the current queue
    behind both; it may be called from several queues."""
    __slots__ = ("works", "ev")

    def __init__(self, works, ev):
        self.works = works
        self.ev = ev

    def wait(self):
        if self.ev is not None:
            self.ev.wait()
        for w in self.works:
            w.wait()


class DistributedLML(object):
    """Evaluate the LML data term of a GP whose K_tot is partitioned over the ranks of ``group``.

    ``X`` (N, D) float64 and ``n`` (N, D) integer derivative orders are replicated on every rank.
    ``fit(kernel_id, params, y, err_y, ...)`` returns ``(ll_data, logdet_half)`` on every rank and
    raises ``numpy.linalg.LinAlgError`` if K_tot is not positive definite.

    ``exchange``: ``"bcast"`` or ``"scatter_gather"`` (panels of at least ``sag_min_bytes`` whose row count divides by
    the world size; smaller ones are broadcast); may be changed between ``fit`` calls (bench.py times both).  Several plans
    (e.g. of different ``nb``) may share one ``ops`` object.
    """
    NBUF = 4

    def __init__(self, X, n, nb=512, group=None, ops=None, device=None, lookahead=True, layout=None,
                 schedule="bcast", exchange="bcast", sag_min_bytes=8 << 20, owner_first=None, inv_trsm=True,
                 inv_min_rows=8192, compiled=None):
        if nb <= 0 or nb % 128:
            raise ValueError("nb must be a positive multiple of 128")
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.grank = self.rank                  # position in the process group (``layout`` does not change it)
        if layout is not None:
            # (rank, world) of the block-cyclic layout given explicitly: a subclass then supplies ``_bcast`` /
            # ``_allreduce`` itself (scratch/sim_ranks.py replays one rank's schedule of an 8-rank job on one GPU)
            self.rank, self.world = layout
        if ops is None:
            ops = HipPanelOps(0 if device is None else device)
        self.ops = ops
        self.device = getattr(ops, "device", torch.device("cpu"))
        # GPT_DIST_FORCE_COLLECTIVES=1 issues the (self-)broadcasts and all-reduces even with a single rank, so the
        # RCCL call pattern can be exercised on a 1-GPU box (tests/test_gpu_a_dist_processes.py).
        self.force_collectives = bool(int(os.environ.get("GPT_DIST_FORCE_COLLECTIVES", "0"))) and dist.is_initialized()
        self.lookahead = bool(lookahead)
        if schedule != "bcast" or exchange not in ("bcast", "scatter_gather"):
            raise ValueError("schedule must be 'bcast' (whole panels), exchange 'bcast' or 'scatter_gather'")
        self.schedule = schedule
        self.exchange = exchange
        self.sag_min_bytes = int(sag_min_bytes)
        # owner_first: the owner of panel k+1 starts its own trailing update of step k only after it has produced panel
        # k+1.  Everybody else is waiting for that panel, nobody for the owner's update; and a panel kernel that has to
        # share the chip with a trailing update already in flight gets a third of it, whatever the stream priority
        # (N=32768, 8 ranks replayed on one GPU: 3.0 ms from first to last chunk of an early panel, against
        # ~0.8 ms when it has the chip).  Pointless with one or two ranks (the owner is always the same / every
        # other step), on by default from three.
        self.owner_first = (self.world > 2) if owner_first is None else bool(owner_first)
        # inv_trsm: the rows of a panel below its diagonal block are solved as ONE GEMM against the
        # explicit inverse of the factored diagonal block (computed once per panel, off the chain) instead of by
        # substitution in four 128-column leaves: twice the flops at several times the rate for tall chunks.
        self.inv_trsm = bool(inv_trsm)
        self.inv_min_rows = int(inv_min_rows)       # whole-panel schedule: panels of at least this many rows take that route
        self._inv_panel = False
        # RCCL runs the collectives of a communicator in issue order on its stream; other backends need explicit waits
        self._stream_ordered = dist.is_initialized() and dist.get_backend(group) == "nccl"
        X = np.ascontiguousarray(X, dtype=np.float64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        self.N, self.D = X.shape
        self._n_maxsum = int(n.sum(axis=1).max()) if n.size else 0
        self.nb = nb
        self.NP = (self.N + 1 + nb - 1) // nb * nb
        self.nblk = self.NP // nb
        self.my_blocks = [J for J in range(self.nblk) if J % self.world == self.rank]
        self.nloc = len(self.my_blocks)
        dev = self.device
        self.X = torch.from_numpy(X).to(dev)
        self.n = torch.from_numpy(n).to(dev)
        self.A = torch.empty((self.NP, max(self.nloc, 1) * nb), dtype=torch.float64, device=dev)
        # panel buffers: panel k is read by the main queue's updates while panel k+1 is staged / received and earlier
        # ones may still be in use by updates that have not drained (the slowest rank sets the pace of the exchanges)
        self.P = [torch.empty((self.NP, nb), dtype=torch.float64, device=dev) for _ in range(self.NBUF)]
        self.S = torch.empty((self.NP, nb), dtype=torch.float64, device=dev)     # staged column below the head (inv_trsm)
        self.Winv = torch.empty((nb, nb), dtype=torch.float64, device=dev)       # L_kk^-1 of the panel being produced
        self.invd = torch.empty(((nb // 128) * 9216,), dtype=torch.float64, device=dev)     # GPT_WS_BLOCK per 128 columns
        self.info = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.y = torch.empty((self.NP,), dtype=torch.float64, device=dev)
        self.err = torch.zeros((self.NP,), dtype=torch.float64, device=dev)
        self.red = torch.zeros((3,), dtype=torch.float64, device=dev)        # sum(log L_ii) part, z.z part, info
        # pinned staging for the per-fit host traffic (y | err_y up, three scalars down): page-locked by the library's own
        # allocator (gpt_host_alloc) and only WRAPPED as tensors -- tensors from torch's pinned-memory allocator that live
        # until interpreter exit crashed the process at teardown ("pure virtual method called", after every result was out)
        on_gpu = dev.type == "cuda"
        self._h_in = torch.from_numpy(_lib.pinned_empty((2, self.NP), min_bytes=0)) if on_gpu else None
        self._h_out = torch.from_numpy(_lib.pinned_empty((3,), min_bytes=0)) if on_gpu else None
        self.timings = {}
        self.trace = False          # record the device-timeline position of every step (timings["steps_ms"])
        # compiled: how an evaluation is driven.  "native": the step loop recorded once, replayed by gpt_plan_run (C loop, RCCL
        # called from the library); "python": the same op list through the Python interpreter (any ops, torch.distributed);
        # False: the step loop itself issues every operation (what rounds 1-5 did; still what a traced evaluation does).
        # None: "native" with the product ops on ONE rank (no communicator of the plan's own is involved beyond a 1-rank one), else
        # "python".  The native replay on several ranks -- RCCL called from the library on the plan's own communicator -- has never
        # run on more than one GPU (this build had none): it is opt-in (compiled="native", GPT_DIST_COMPILED=native; bench.py tries it
        # as a leg under its watchdog) until it has.  GPT_DIST_COMPILED=0 / native / python overrides.
        env = os.environ.get("GPT_DIST_COMPILED")
        if env is not None:
            compiled = False if env == "0" else env
        if compiled is None:
            one_rank = not dist.is_initialized() or dist.get_world_size(group) == 1
            compiled = "native" if (isinstance(ops, HipPanelOps) and one_rank and layout is None) else "python"
        if compiled not in (False, "native", "python"):
            raise ValueError("compiled must be None, False, 'native' or 'python'")
        if compiled == "native" and not isinstance(ops, HipPanelOps):
            raise ValueError("compiled='native' needs the product ops (HipPanelOps)")
        self.compiled = compiled
        self._plans = {}
        self._rec = None

    # ------------------------------------------------------------------------------------------
    def _assemble(self, kernel_id, params, noise_var, diag_add):
        """Each rank builds its block columns of K_tot (lower part) plus the padding / augmented row."""
        N, nb, A = self.N, self.nb, self.A
        ld = A.stride(0)
        for lj, J in enumerate(self.my_blocks):
            c0, c1 = J * nb, min((J + 1) * nb, N)
            if c0 < N:
                self.ops.kbuild_block(kernel_id, params, self.X, self.n, c0, N, c0, c1, self.err, noise_var, diag_add,
                                      _ptr(A, c0, lj * nb), ld)
            # rows N..NP: augmented row y^T, unit diagonal on the padding, a huge pivot under the augmented row
            self.ops.pad_block(A, lj, c0, nb, N, self.NP, self.y, BIG_PIVOT)

    def _collectives_on(self):
        return self.world > 1 or self.force_collectives

    def _bcast(self, buf, src, async_op=False, group=None):
        if not self._collectives_on():
            return None
        group = self.group if group is None else group
        gsrc = dist.get_global_rank(group, src) if group is not None else src
        return dist.broadcast(buf, src=gsrc, group=group, async_op=async_op)

    def _use_scatter_gather(self, buf):
        return (self.exchange == "scatter_gather" and buf.shape[0] % self.world == 0
                and buf.numel() * buf.element_size() >= self.sag_min_bytes)

    def _exchange(self, buf, src, group=None, tag=None):
        """Start moving the contiguous rows ``buf`` from rank ``src`` to everyone; returns the list of handles to wait for
        (empty when no collective is needed).  scatter + all-gather: the root's links each carry 1/world of the rows, then
        every link carries the all-gather; a broadcast moves the whole chunk along one path.  While the step loop is being
        RECORDED (compiled schedules) the exchange becomes ops of the list.  ``tag`` = (panel, first row) is not used here
        (scratch/sim_ranks.py overrides this method and needs to know what is being moved)."""
        if not self._collectives_on():
            return []
        if self._rec is not None:
            return self._rec.exchange(buf, src, self.world, self.grank, self._use_scatter_gather(buf))
        return self._exchange_now(buf, src, self._use_scatter_gather(buf), group)

    def _exchange_now(self, buf, src, scatter_gather, group=None):
        W = self.world
        if scatter_gather:
            group = self.group if group is None else group
            gsrc = dist.get_global_rank(group, src) if group is not None else src
            c = buf.shape[0] // W
            pieces = [buf[r * c:(r + 1) * c] for r in range(W)]
            mine = pieces[self.grank]
            w1 = dist.scatter(mine, scatter_list=pieces if self.grank == src else None, src=gsrc, group=group,
                              async_op=True)
            if not self._stream_ordered:
                w1.wait()      # (gloo runs asynchronous operations on a thread pool, in no particular order)
            w2 = dist.all_gather_into_tensor(buf, mine, group=group, async_op=True)      # in place
            return [w1, w2]
        return [self._bcast(buf, src, async_op=True, group=group)]

    def _plan_collective(self, opcode, buf, src):
        """(Python interpreter of a compiled schedule) issue the recorded exchange; returns the work handles."""
        return self._exchange_now(buf, src, scatter_gather=(opcode == OP_SCATTER))

    def _allreduce(self, t, op):
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)

    def _stage_panel(self, k, buf, allow_inv=True):
        """Owner side: copy block column k (every update before panel k-1 applied) into the contiguous panel
        buffer -- or, for a tall panel with ``inv_trsm``, into the scratch column S, from where ``_factor_staged``
        produces the panel (``_staged`` names the one in use).  Issued *before* waiting for panel k-1 so the strided
        copy is off the critical chain."""
        nb = self.nb
        lk = k // self.world
        m = self.NP - k * nb
        self._inv_panel = bool(allow_inv and self.inv_trsm and m >= self.inv_min_rows and m > nb)
        dst = self.S if self._inv_panel else buf
        self.ops.copy2d(dst[:m], self.A[k * nb:, lk * nb:(lk + 1) * nb])

    def _staged(self, buf):
        return self.S if self._inv_panel else buf

    def _factor_staged(self, k, buf):
        """Owner side: factor the staged panel into the broadcast buffer.  L is never copied back: the
        local matrix is only a work area, the scalars the LML needs are accumulated here from the panel.
        Tall panels (``inv_trsm``): only the diagonal block goes through the panel factorisation; the rows below are
        ONE GEMM of the staged rows against the explicit inverse of the factored block (0.12 ms for the inverse, then
        31k x 512 rows in 0.35 ms instead of 0.86 by four 128-column substitution leaves and three narrow updates)."""
        nb, N, ops = self.nb, self.N, self.ops
        m = self.NP - k * nb
        if self._inv_panel:
            ops.copy2d(buf[:nb], self.S[:nb])
            ops.potrf_panel(nb, nb, buf.data_ptr(), nb, self.invd, self.info, k * nb)
            ops.trinv(nb, buf.data_ptr(), nb, self.invd, self.Winv.data_ptr(), nb)
            ops.gemm_nt(m - nb, nb, nb, 1.0, _ptr(self.S, nb, 0), nb, self.Winv.data_ptr(), nb, 0.0, _ptr(buf, nb, 0), nb, 0,
                        q="panel")
        else:
            ops.potrf_panel(m, nb, buf.data_ptr(), nb, self.invd, self.info, k * nb)
        self._factored.append((k, buf))

    def _accumulate_scalars(self):
        """sum(log L_ii) over i < N and z.z (augmented row N) from the panels this rank factored since the
        last call; runs after the broadcast of the panel has been enqueued."""
        nb, N = self.nb, self.N
        for k, buf in self._factored:
            w = min(nb, N - k * nb)
            if w <= 0:
                continue
            self.ops.panel_scalars(buf, w, N - k * nb, self.red)
        self._factored = []

    def _update_block(self, k, J, buf, C=None, ldc=None, q="main"):
        """A[J*nb:, block J] -= P_k[rows of J..] * P_k[rows of block J]^T  (lower trapezoid); with ``C`` given the
        target is a staged panel buffer instead of the local matrix."""
        nb, A = self.nb, self.A
        lj = J // self.world
        mJ = self.NP - J * nb
        off = (J - k) * nb
        if C is None:
            C, ldc = _ptr(A, J * nb, lj * nb), A.stride(0)
        self.ops.gemm_nt(mJ, nb, nb, -1.0, _ptr(buf, off, 0), nb, _ptr(buf, off, 0), nb, 1.0, C, ldc, 1, q=q)

    def _update_blocks(self, k, Js, buf):
        """Panel k applied to the owned block columns ``Js`` (ascending, adjacent in local storage, ``world`` apart in
        the matrix) in one staircase launch: segment s is block column Js[0] + s * world."""
        nb, A, W = self.nb, self.A, self.world
        J0 = Js[0]
        assert all(J == J0 + s * W for s, J in enumerate(Js))
        off = (J0 - k) * nb
        self.ops.gemm_nt_stair(self.NP - J0 * nb, len(Js), nb, nb, -1.0, _ptr(buf, off, 0), nb, _ptr(buf, off, 0), nb,
                               W * nb, W * nb, 1.0, _ptr(A, J0 * nb, (J0 // W) * nb), A.stride(0), q="main")

    def _upload(self, y, err_y):
        """Per-evaluation host traffic and resets, on the main queue: y | err_y up, info and the scalar accumulators to zero."""
        ops, N = self.ops, self.N
        y = np.ascontiguousarray(y, dtype=np.float64)
        err_y = np.array(np.broadcast_to(err_y, (N,)), dtype=np.float64)
        with ops.queue("main"):
            # y | err_y go up from a pinned staging tensor, asynchronously on the main queue (a pageable source makes the copy
            # synchronous and costs a staging copy inside the runtime); the scalar accumulators are persistent tensors
            if self._h_in is not None:
                self._h_in[0, :N].copy_(torch.from_numpy(y))
                self._h_in[1, :N].copy_(torch.from_numpy(err_y))
                self.y[:N].copy_(self._h_in[0, :N], non_blocking=True)
                self.err[:N].copy_(self._h_in[1, :N], non_blocking=True)
            else:
                self.y[:N] = torch.from_numpy(y)
                self.err[:N] = torch.from_numpy(err_y)
            self.info.zero_()
            self.red.zero_()

    def _begin(self, kernel_id, params, noise_var, diag_factor):
        """Head of the schedule (recorded in a compiled plan): build the local block columns on the main queue."""
        ops = self.ops
        with ops.queue("main"):
            self._t0 = ops.new_timing_event() if (self.trace and self._rec is None) else None
            if self._t0 is not None:
                self._t0.record()
            self._assemble(kernel_id, params, noise_var, diag_factor * sys.float_info.epsilon)
            ev_asm = ops.new_event()
            ev_asm.record()
        self._factored = []
        self._marks = []
        return ev_asm

    def _mark(self, k, tag):
        """(trace) device-timeline stamp on the current (main) queue: panel k has arrived / has been applied."""
        if self._t0 is not None:
            e = self.ops.new_timing_event()
            e.record()
            self._marks.append((k, tag, e))

    def _finish(self, t_host0):
        """Common tail: the three scalars (sum(log L_ii) over i < N, z.z from the augmented row, info) and ll."""
        ops, N, world = self.ops, self.N, self.world
        self.timings["host_enqueue_s"] = time.perf_counter() - t_host0      # the host ran this far ahead of the GPU
        with ops.queue("panel"):
            red = self.red
            red[2] = self.info.to(torch.float64)[0]
            if self._collectives_on():
                # info: non-zero on the owner of the failing panel only; max picks it up
                info_t = red[2:3].clone()
                self._allreduce(red[:2], "sum")
                self._allreduce(info_t, "max")
                red[2] = info_t[0]
            if self._h_out is not None:
                self._h_out.copy_(red, non_blocking=True)         # pinned: the wait below is the only synchronisation
        ops.synchronize()
        logdet_half, zz, info = (float(v) for v in (self._h_out if self._h_out is not None else red))
        if self._t0 is not None:
            self.timings["steps_ms"] = [(k, tag, self._t0.elapsed_ms(e)) for k, tag, e in self._marks]
        if info != 0 and info <= N:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % int(info))
        ll_data = -0.5 * zz - logdet_half - 0.5 * N * math.log(2.0 * math.pi)
        if info != 0 or not math.isfinite(ll_data):
            # like the single-GPU path (api.hip factor_and_ll): a failure in the augmented row -- z.z overflowed, or y / K
            # held non-finite values -- is an error, not a NaN handed to the optimiser
            raise np.linalg.LinAlgError("factorisation failed in the augmented row (non-finite y or K_tot?)")
        return ll_data, logdet_half

    def fit(self, kernel_id, params, y, err_y, noise_var=0.0, diag_factor=1e2):
        """One LML evaluation; returns ``(ll_data, logdet_half)`` on every rank."""
        # the derivative-order limits gpt_fit / gpt_fit_sum check on the host (the device API takes what it is given)
        if kernel_id == _lib.KERNEL_M52 and self._n_maxsum > 1:
            raise ValueError("Matern52Kernel only supports 0th and 1st order derivatives")      # ref matern.py:545-546
        if kernel_id in (_lib.KERNEL_RQ, _lib.KERNEL_MATERN) and 2 * self._n_maxsum > 16:
            raise ValueError("RationalQuadratic / Matern kernel: derivative orders of a pair sum to %d, the device builder "
                             "supports 16" % (2 * self._n_maxsum))
        return self._fit_bcast(kernel_id, params, y, err_y, noise_var, diag_factor)

    # ------------------------------------------------------------------------------------------
    def _fit_bcast(self, kernel_id, params, y, err_y, noise_var, diag_factor):
        """Whole-panel schedule.  Per step k (panel k = block column k of L, contiguous in P[k % NBUF] on every rank):
          panel queue: [owner of k+1: stage block column k+1 into P[(k+1) % NBUF]] -> panel k has arrived ->
                       [owner: apply panel k to the staged column, factor it] -> start exchange k+1 (async)
          main queue : panel k has arrived -> apply it to the owned block columns right of k+1, the one that is staged
                       next (k+2) first.
        Edges: "urgent" (column k+2 is up to date with panel k) main -> panel, "done" (step k no longer reads its
        buffer) main -> panel before that buffer is staged / received into again, "arrived" panel -> main."""
        t_host0 = time.perf_counter()
        self._upload(y, err_y)
        if self.compiled and not self.trace:
            diag_add = diag_factor * sys.float_info.epsilon
            key = (self.compiled, self.exchange, self.lookahead, self.owner_first, self.inv_trsm, self.inv_min_rows, self.sag_min_bytes,
                   self._collectives_on())
            plan = self._plans.get(key)
            if plan is None:
                # walk the step loop once against the recorder (nothing runs), then build the executor
                real, rec = self.ops, PlanRecorder(self.device)
                self.ops = self._rec = rec
                try:
                    self._schedule(None, None, 0.0, 0.0)
                finally:
                    self.ops, self._rec = real, None
                plan = CompiledPlan(rec)
                if self.compiled == "native":
                    plan.build_native(real, self.X, self.n, self.err,
                                      [(0, self.world, self.grank, self.group)] if self._collectives_on() else [])
                self._plans[key] = plan
                self.timings["plan_ops"] = len(plan.ops)
            if self.compiled == "native":
                self.timings["native_enqueue_ms"] = plan.run_native(kernel_id, params, noise_var, diag_add)
            else:
                plan.run_python(self.ops, self, kernel_id, params, noise_var, diag_add)
            return self._finish(t_host0)
        self._schedule(kernel_id, params, noise_var, diag_factor)
        return self._finish(t_host0)

    def _schedule(self, kernel_id, params, noise_var, diag_factor):
        """The step loop of the whole-panel schedule (see _fit_bcast): issues the operations -- or, against a PlanRecorder, lists
        them."""
        ops = self.ops
        nb, NP, world, rank, NBUF = self.nb, self.NP, self.world, self.rank, self.NBUF
        nblk = self.nblk
        owner = lambda J: J % world == rank
        ev_asm = self._begin(kernel_id, params, noise_var, diag_factor)
        ev_urg, ev_done = {}, {}

        def wait_all(ws):
            for w in ws:
                w.wait()

        with ops.queue("panel"):
            ev_asm.wait()
            if owner(0):
                self._stage_panel(0, self.P[0])
                self._factor_staged(0, self.P[0])
            pending = self._exchange(self.P[0][:NP], 0, tag=(0, 0))
            self._accumulate_scalars()

        for k in range(nblk):
            buf = self.P[k % NBUF]
            nxt = k + 1
            nbuf = self.P[nxt % NBUF]
            own_next = nxt < nblk and owner(nxt)
            la = self.lookahead and nxt < nblk
            with ops.queue("panel"):
                if la:
                    if nxt - NBUF in ev_done:
                        ev_done.pop(nxt - NBUF).wait()     # nbuf held panel k+1-NBUF
                    if own_next:
                        if k - 1 in ev_urg:
                            ev_urg.pop(k - 1).wait()       # column k+1 is up to date with panel k-1
                        self._stage_panel(nxt, nbuf)       # before the wait: overlaps the tail of exchange k
                wait_all(pending)
                pending = []
                ev_arr = ops.new_event()
                ev_arr.record()
                ev_own = None
                if la:
                    if own_next:
                        self._update_block(k, nxt, buf, self._staged(nbuf).data_ptr(), nb, q="panel")
                        self._factor_staged(nxt, nbuf)
                        if self.owner_first:
                            ev_own = ops.new_event()
                            ev_own.record()
                    pending = self._exchange(nbuf[:NP - nxt * nb], nxt % world, tag=(nxt, 0))
                    self._accumulate_scalars()
            with ops.queue("main"):
                ev_arr.wait()
                if ev_own is not None:
                    ev_own.wait()                          # own panel first (see __init__)
                self._mark(k, "arrived")
                mine = [J for J in self.my_blocks if J > k and not (la and J == nxt)]
                urgent = k + 2
                if la and urgent in mine:
                    self._update_block(k, urgent, buf)
                    mine.remove(urgent)
                    ev_urg[k] = ops.new_event()
                    ev_urg[k].record()
                if mine:
                    self._update_blocks(k, mine, buf)
                ev_done[k] = ops.new_event()
                ev_done[k].record()
                self._mark(k, "applied")
            if nxt < nblk and not la:
                # no look-ahead: the next panel is factored only after every update of this step
                with ops.queue("panel"):
                    ev_done.pop(k).wait()
                    if nxt - NBUF in ev_done:
                        ev_done.pop(nxt - NBUF).wait()
                    if own_next:
                        self._stage_panel(nxt, nbuf)
                        self._factor_staged(nxt, nbuf)
                    pending = self._exchange(nbuf[:NP - nxt * nb], nxt % world, tag=(nxt, 0))
                    self._accumulate_scalars()


# ======================================================================================================
# 2-D block-cyclic layout over a P_r x P_c process grid
# ======================================================================================================
def _ceil_div(a, b):
    return -((-a) // b)


class GridLML(object):
    """The same LML evaluation with K_tot spread over a ``P_r x P_c`` PROCESS GRID.

    Why: in the 1-D layout of :class:`DistributedLML` the serial work per panel -- apply panel k to block column k+1, factor
    its diagonal block, solve the rows under it -- sits on ONE GPU by construction: 0.6 ms x 64 panels at N = 32768 on 8 ranks,
    as long as the whole budget.  Here block (I, J) of the ``nb``-blocked matrix lives on grid position
    ``(I % P_r, J % P_c)`` (rank ``pr * P_c + pc``), so the column update and the solve of a panel are split over the ``P_r``
    ranks of a process column, and only the ``nb x nb`` diagonal block is serial.  K is still built locally (X replicated).

    Panel k = block column k of L; process column ``pc_k = k % P_c``, diagonal owner ``(k % P_r, pc_k)``.  Three in-order queues
    per rank, and what each carries is chosen so that the serial CHAIN never queues behind BULK work:
      CHAIN queue ("panel", high priority):
        1. diagonal owner: ``A[k][k] -= H(k-1) H(k-1)^T``, factor it in place (``gpt_dev_potrf_panel``), ``W = L_kk^-1``
           (``gpt_dev_trinv``), broadcast W down its process column;
        2. the HEAD block ``H(k) = L[k+1][k]`` on its holder ``((k+1) % P_r, pc_k)``: ``A[k+1][k] -= R0(k-1) H(k-1)^T`` (R0:
           below), ``H(k) = A[k+1][k] W^T``, broadcast to everybody.  Of panel k-1 the chain reads exactly two blocks,
           ``H(k-1) = L[k][k-1]`` and ``R0(k-1) = L[k+1][k-1]``, both sent early on communicators of their own; per panel it is
           one 512 x 512 factorisation + inverse, three 512^3 products and two small broadcasts long, whatever the panel's height.
      BULK queue ("recv"):
        3. look-ahead: the ranks of process column ``pc_k`` apply panel k-1 to their rows I >= k+1 of block column k
           (``A[I][k] -= R(k-1)[I] H(k-1)^T``; the head block is the chain's);
        4. every rank ``(pr, pc_k)`` solves its rows I >= k+2 of the panel as GEMMs against W -- the EARLY block
           ``R0(k) = L[k+2][k]`` first where its process row has it -- and broadcasts them along its process row: every rank then
           holds the rows ``R`` of panel k that match its block rows;
        5. the columns ``C`` of panel k that match a rank's block columns J >= k+2 are blocks of ``R`` on the ranks of process
           row ``J % P_r``: exchanged inside each process column (one broadcast per contributing process row: one for
           ``P_r | P_c``, ``P_r / gcd`` in general).
      MAIN queue (CU-masked):
        6. ``A[I][J] -= R[I] C[J]^T`` for the rank's blocks with I >= J >= k+2, one launch (``gpt_dev_gemm_nt_gridstair``), the
           column the next look-ahead touches first ("urgent").
    (Round 4's first form had 1-5 on ONE queue: the head block of panel k+1 then waited behind the look-ahead update of step k,
    which waits for the bulk of panel k to arrive -- modelled at 73-185 ms for C4 on 8 ranks, the chain advancing at the pace of
    the bulk transfers.)
    ``z = L^-1 y`` rides along as the augmented row N and is summed where its pieces come to rest; one
    all-reduce of (sum log L_ii, z.z, info) ends the evaluation.  ``fit`` returns the same ``(ll_data, logdet_half)`` on every
    rank.  ``grid = (1, W)`` is the 1-D block-column layout, ``(W, 1)`` a block-row layout.  ``compiled`` as for
    :class:`DistributedLML`: the step loop below is recorded once per (look-ahead) as an op list and replayed by ``gpt_plan_run``
    ("native": five RCCL channels per rank -- bulk rows / early block on the process row, inverse / column exchange on the process
    column, head blocks on the whole grid) or by the Python interpreter of the list ("python"); False: the loop issues every
    operation itself.  UNMEASURED ON MORE THAN ONE GPU:
    gloo worlds 2-8 on CPU (tests/test_dist_gloo.py), ranks sharing one GPU through the product ops, and the modelled 8-rank
    time of scratch/sim_model.py are what exists.
    """
    NBUF = 4

    # communication channels of a compiled schedule (queue names of the op list): what moves on which communicator
    CHANNEL = {"H": "comm", "R": "comm1", "R0": "comm2", "W": "comm3", "C": "comm4"}

    def __init__(self, X, n, grid, nb=512, group=None, ops=None, device=None, lookahead=True, layout=None, compiled=None):
        if nb <= 0 or nb % 128:
            raise ValueError("nb must be a positive multiple of 128")
        self.Pr, self.Pc = int(grid[0]), int(grid[1])
        if self.Pr < 1 or self.Pc < 1:
            raise ValueError("grid must be (P_r, P_c) with both >= 1")
        self.group = group
        inited = dist.is_initialized()
        self.rank = dist.get_rank(group) if inited else 0
        self.world = dist.get_world_size(group) if inited else 1
        self._model = layout is not None
        if layout is not None:
            # position in the grid given explicitly: a subclass supplies ``_xbcast`` / ``_allreduce`` (scratch/sim_model.py
            # replays one rank of an 8-rank job on one GPU)
            self.rank, self.world = int(layout), self.Pr * self.Pc
        if self.world != self.Pr * self.Pc:
            raise ValueError("grid %d x %d needs %d ranks, the group has %d" % (self.Pr, self.Pc, self.Pr * self.Pc, self.world))
        self.pr, self.pc = divmod(self.rank, self.Pc)
        if ops is None:
            ops = HipPanelOps(0 if device is None else device)
        self.ops = ops
        self.device = getattr(ops, "device", torch.device("cpu"))
        self.force_collectives = bool(int(os.environ.get("GPT_DIST_FORCE_COLLECTIVES", "0"))) and inited
        self.lookahead = bool(lookahead)
        # communicators: one per process row (panel rows), two per process column (the inverse of the diagonal block / the
        # column exchange: the latter must never queue behind a diagonal block that is still being factored), the whole grid
        # for the head blocks and the final reduction.  Every rank creates every group, in the same order.
        self.g_row = self.g_row0 = self.g_colw = self.g_colx = None
        self._ranks = list(range(self.world))
        if inited and layout is None:
            self._ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            if self.world > 1 or self.force_collectives:
                for r in range(self.Pr):
                    members = [self._ranks[r * self.Pc + c] for c in range(self.Pc)]
                    g, g0 = dist.new_group(ranks=members), dist.new_group(ranks=members)
                    if r == self.pr:
                        self.g_row, self.g_row0 = g, g0
                for c in range(self.Pc):
                    members = [self._ranks[r * self.Pc + c] for r in range(self.Pr)]
                    gw, gx = dist.new_group(ranks=members), dist.new_group(ranks=members)
                    if c == self.pc:
                        self.g_colw, self.g_colx = gw, gx
        X = np.ascontiguousarray(X, dtype=np.float64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        self.N, self.D = X.shape
        self._n_maxsum = int(n.sum(axis=1).max()) if n.size else 0
        self.nb = nb
        self.NP = (self.N + 1 + nb - 1) // nb * nb
        self.nblk = nblk = self.NP // nb
        self.my_rows = [I for I in range(nblk) if I % self.Pr == self.pr]
        self.my_cols = [J for J in range(nblk) if J % self.Pc == self.pc]
        self.my_blocks = [(I, J) for J in self.my_cols for I in self.my_rows if I >= J]
        self.nlr, self.nlc = len(self.my_rows), len(self.my_cols)
        self.lcm = self.Pr * self.Pc // math.gcd(self.Pr, self.Pc)
        dev = self.device
        self.X = torch.from_numpy(X).to(dev)
        self.n = torch.from_numpy(n).to(dev)
        # the rank's own block rows of X / n, gathered (zero rows stand for the padding beyond N)
        Xr = np.zeros((max(self.nlr, 1) * nb, self.D))
        nr = np.zeros((max(self.nlr, 1) * nb, self.D), dtype=np.int32)
        for li, I in enumerate(self.my_rows):
            r0, r1 = I * nb, min((I + 1) * nb, self.N)
            if r0 < r1:
                Xr[li * nb:li * nb + r1 - r0] = X[r0:r1]
                nr[li * nb:li * nb + r1 - r0] = n[r0:r1]
        self.Xr = torch.from_numpy(Xr).to(dev)
        self.nr = torch.from_numpy(nr).to(dev)
        rows, cols = max(self.nlr, 1) * nb, max(self.nlc, 1) * nb
        self.A = torch.empty((rows, cols), dtype=torch.float64, device=dev)
        f64 = dict(dtype=torch.float64, device=dev)
        self.R = [torch.empty((rows, nb), **f64) for _ in range(self.NBUF)]       # rows of panel k, by local block row
        self.C = [torch.empty((cols, nb), **f64) for _ in range(self.NBUF)]       # columns of panel k, by local block column
        self.H = [torch.empty((nb, nb), **f64) for _ in range(self.NBUF)]         # head block L[k+1][k]
        self.W = [torch.empty((nb, nb), **f64) for _ in range(self.NBUF)]         # L_kk^-1
        # one contributing process row's share of C, where several rows contribute (lcm / P_c > 1): a buffer per slot and
        # contributing row (a single one would serialise consecutive exchanges on it)
        nq = (self.Pr // math.gcd(self.Pr, self.Pc)) if self.lcm // self.Pc > 1 else 0
        sc = self.lcm // self.Pc
        # sized in whole blocks: one contributing row holds ceil(nlc / sc) of the rank's block columns (rows of
        # ceil(cols / sc) are not a whole number of blocks when nlc % sc != 0)
        self.piece = [[torch.empty((-(-max(self.nlc, 1) // sc) * nb, nb), **f64) for _ in range(nq)] for _ in range(self.NBUF)]
        self.invd = torch.empty(((nb // 128) * 9216,), **f64)
        self.info = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.y = torch.empty((self.NP,), **f64)
        self.err = torch.zeros((self.NP,), **f64)
        self.red = torch.zeros((3,), **f64)
        self.red_b = torch.zeros((3,), **f64)                 # the bulk queue's share of the two ll scalars (its own accumulator)
        on_gpu = dev.type == "cuda"
        self._h_in = torch.from_numpy(_lib.pinned_empty((2, self.NP), min_bytes=0)) if on_gpu else None
        self._h_out = torch.from_numpy(_lib.pinned_empty((3,), min_bytes=0)) if on_gpu else None
        self.timings = {}
        self.trace = False
        # compiled: as for DistributedLML -- "native" (gpt_plan_run replays the recorded op list, RCCL from the library: five
        # communicators per rank), "python" (the same list through the Python interpreter + torch.distributed) or False (the step
        # loop issues every operation; what a traced evaluation and the single-GPU model of scratch/ do).  None: "native" with the
        # product ops on one rank, "python" on several (the native replay has not run on more than one GPU), False for a model.
        env = os.environ.get("GPT_DIST_COMPILED")
        if env is not None and layout is None:
            compiled = False if env == "0" else env
        if compiled is None:
            one_rank = not inited or dist.get_world_size(group) == 1
            compiled = False if layout is not None else ("native" if (isinstance(ops, HipPanelOps) and one_rank) else "python")
        if compiled not in (False, "native", "python"):
            raise ValueError("compiled must be None, False, 'native' or 'python'")
        if compiled == "native" and not isinstance(ops, HipPanelOps):
            raise ValueError("compiled='native' needs the product ops (HipPanelOps)")
        self.compiled = compiled
        self._plans = {}
        self._rec = None

    # ---- index helpers ---------------------------------------------------------------------------
    def li_ge(self, I0):
        """Local index of the rank's first block row I >= I0 (may equal ``nlr``: none)."""
        return min(max(_ceil_div(I0 - self.pr, self.Pr), 0), self.nlr)

    def lj_ge(self, J0):
        return min(max(_ceil_div(J0 - self.pc, self.Pc), 0), self.nlc)

    def _blk(self, I, J):
        """View of the rank's block (I, J)."""
        nb, li, lj = self.nb, I // self.Pr, J // self.Pc
        return self.A[li * nb:(li + 1) * nb, lj * nb:(lj + 1) * nb]

    # ---- communication (overridden by the single-GPU model) --------------------------------------
    def _on(self, size):
        return (size > 1 and self.world > 1) or self.force_collectives

    def _xbcast(self, kind, k, buf, src, group, size):
        """Start the broadcast of the contiguous ``buf`` from grid position ``src`` = (pr, pc) inside ``group`` (``size``
        members); returns the work handles to wait for.  ``kind`` in "W", "H", "R", "C" and the panel index ``k`` name what moves
        (the model needs to know; the product path does not)."""
        if not self._on(size) or buf.numel() == 0:
            return []
        if self._rec is not None:
            # (recording a compiled schedule: the broadcast becomes ops of the list; root = position inside the communicator --
            # process rows are ordered by column, process columns by row, the whole grid row-major)
            root = {"W": src[0], "C": src[0], "R": src[1], "R0": src[1], "H": src[0] * self.Pc + src[1]}[kind]
            return self._rec.bcast(buf, root, self.CHANNEL[kind], (kind, src))
        gsrc = self._ranks[src[0] * self.Pc + src[1]]
        return [dist.broadcast(buf, src=gsrc, group=group, async_op=True)]

    def _plan_collective(self, opcode, buf, what):
        """(Python interpreter of a compiled schedule) issue the recorded broadcast; returns the work handles."""
        kind, src = what
        group = {"H": self.group, "R": self.g_row, "R0": self.g_row0, "W": self.g_colw, "C": self.g_colx}[kind]
        gsrc = self._ranks[src[0] * self.Pc + src[1]]
        return [dist.broadcast(buf, src=gsrc, group=group, async_op=True)]

    def _channels(self):
        """The communicators of a native plan: (channel, ranks, own position, torch group carrying the id), in an order every
        rank shares."""
        if self._model or not dist.is_initialized():
            return []
        out = []
        for kind, size, pos, group in (("H", self.world, self.rank, self.group), ("R", self.Pc, self.pc, self.g_row),
                                       ("R0", self.Pc, self.pc, self.g_row0), ("W", self.Pr, self.pr, self.g_colw),
                                       ("C", self.Pr, self.pr, self.g_colx)):
            if self._on(size):
                out.append((QUEUE_ID[self.CHANNEL[kind]] - 3, size, pos, group))
        return out

    def _allreduce(self, t, op):
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)

    # ---- K build ----------------------------------------------------------------------------------
    def _assemble(self, kernel_id, params, noise_var, diag_add):
        """The rank's blocks (I, J), I >= J: one rectangle of the fused builder per local block column for the blocks below the
        diagonal, the diagonal blocks (with the diagonal loading of ref gaussian_process.py:1447-1451) one by one, then the
        padding / augmented rows where the rank holds the last block row."""
        N, nb, A, ops = self.N, self.nb, self.A, self.ops
        ld = A.stride(0)
        last = self.nblk - 1
        for lj, J in enumerate(self.my_cols):
            c0, c1 = J * nb, min((J + 1) * nb, N)
            if c0 < N:
                li_s = self.li_ge(J + 1)
                if li_s < self.nlr:
                    ops.kbuild_rect(kernel_id, params, self.Xr, self.nr, li_s * nb, self.nlr * nb, self.X, self.n, c0, c1,
                                    _ptr(A, li_s * nb, lj * nb), ld)
                if J % self.Pr == self.pr:
                    ops.kbuild_block(kernel_id, params, self.X, self.n, c0, c1, c0, c1, self.err, noise_var, diag_add,
                                     _ptr(A, (J // self.Pr) * nb, lj * nb), ld)
            if last % self.Pr == self.pr:
                ops.pad_block(A, lj, c0, nb, N, self.NP, self.y, BIG_PIVOT, row_shift=(last // self.Pr - last) * nb)

    # ---- one evaluation -------------------------------------------------------------------------
    def fit(self, kernel_id, params, y, err_y, noise_var=0.0, diag_factor=1e2):
        """One LML evaluation; returns ``(ll_data, logdet_half)`` on every rank (``numpy.linalg.LinAlgError`` on every rank
        if K_tot is not positive definite)."""
        if kernel_id == _lib.KERNEL_M52 and self._n_maxsum > 1:
            raise ValueError("Matern52Kernel only supports 0th and 1st order derivatives")      # ref matern.py:545-546
        if kernel_id in (_lib.KERNEL_RQ, _lib.KERNEL_MATERN) and 2 * self._n_maxsum > 16:
            raise ValueError("RationalQuadratic / Matern kernel: derivative orders of a pair sum to %d, the device builder "
                             "supports 16" % (2 * self._n_maxsum))
        t_host0 = time.perf_counter()
        self._upload(y, err_y)
        if self.compiled and not self.trace:
            key = (self.compiled, self.lookahead)
            plan = self._plans.get(key)
            if plan is None:
                # walk the step loop once against the recorder (nothing runs), then build the executor
                real, rec = self.ops, PlanRecorder(self.device)
                self.ops = self._rec = rec
                try:
                    self._schedule(None, None, 0.0, 0.0)
                finally:
                    self.ops, self._rec = real, None
                plan = CompiledPlan(rec)
                if self.compiled == "native":
                    plan.build_native(real, self.X, self.n, self.err, self._channels())
                self._plans[key] = plan
                self.timings["plan_ops"] = len(plan.ops)
            diag_add = diag_factor * sys.float_info.epsilon
            if self.compiled == "native":
                self.timings["native_enqueue_ms"] = plan.run_native(kernel_id, params, noise_var, diag_add)
            else:
                plan.run_python(self.ops, self, kernel_id, params, noise_var, diag_add)
            return self._finish(t_host0)
        self._schedule(kernel_id, params, noise_var, diag_factor)
        return self._finish(t_host0)

    def _upload(self, y, err_y):
        """Per-evaluation host traffic and resets, on the main queue: y | err_y up, info and the scalar accumulators to zero."""
        ops, N = self.ops, self.N
        y = np.ascontiguousarray(y, dtype=np.float64)
        err_y = np.array(np.broadcast_to(err_y, (N,)), dtype=np.float64)
        with ops.queue("main"):
            if self._h_in is not None:
                self._h_in[0, :N].copy_(torch.from_numpy(y))
                self._h_in[1, :N].copy_(torch.from_numpy(err_y))
                self.y[:N].copy_(self._h_in[0, :N], non_blocking=True)
                self.err[:N].copy_(self._h_in[1, :N], non_blocking=True)
            else:
                self.y[:N] = torch.from_numpy(y)
                self.err[:N] = torch.from_numpy(err_y)
            self.info.zero_()
            self.red.zero_()
            self.red_b.zero_()          # (HERE, on the main queue in front of ev_asm: zeroed on the default stream it raced with the
                                        #  bulk queue's accumulation -- one evaluation in 72 lost part of z.z under queue jitter)

    def _schedule(self, kernel_id, params, noise_var, diag_factor):
        """The step loop (class docstring): issues the operations -- or, against a PlanRecorder, lists them."""
        ops, N, nb, NP, nblk, NBUF = self.ops, self.N, self.nb, self.NP, self.nblk, self.NBUF
        Pr, Pc, pr, pc = self.Pr, self.Pc, self.pr, self.pc
        A, ld = self.A, self.A.stride(0)
        with ops.queue("main"):
            self._t0 = ops.new_timing_event() if (self.trace and self._rec is None) else None
            if self._t0 is not None:
                self._t0.record()
            self._assemble(kernel_id, params, noise_var, diag_factor * sys.float_info.epsilon)
            ev_asm = ops.new_event()
            ev_asm.record()
        self._marks = []
        for q in ("panel", "recv"):
            with ops.queue(q):
                ev_asm.wait()
        last = nblk - 1
        zrow_local = (last // Pr) * nb + (N - last * nb)       # local row of the augmented row on the ranks of its process row
        CH, BK = "panel", "recv"                               # the CHAIN queue and the BULK queue (see the class docstring)
        ev_urg, ev_done, ev_ch, ev_bk = {}, {}, {}, {}
        arr_W, arr_H, arr_R0, arr_R, arr_C = {}, {}, {}, {}, {}
        self._late = []

        def reuse(p):
            """Slot p % NBUF is about to be written for panel p by the current queue: whoever still read panel p - NBUF from it
            -- the main queue's update, the chain queue, the bulk queue -- must be through."""
            for d in (ev_done, ev_ch, ev_bk):
                e_ = d.get(p - NBUF)
                if e_ is not None:
                    e_.wait()

        def wait_all(arrs):
            for a_ in arrs:
                if a_ is not None:
                    a_.wait()

        def chain(k):
            """Chain queue, panel p = k + 1 (k = -1: panel 0): the diagonal block and the head block L[p+1][p] -- what the NEXT
            diagonal block waits for.  Needs of panel k only the two early blocks H(k) = L[k+1][k] and R0(k) = L[k+2][k]."""
            p = k + 1
            s, s1 = k % NBUF, p % NBUF
            pcp = p % Pc
            with ops.queue(CH):
                reuse(p)
                if k >= 0 and not self.lookahead:
                    ev_done[k].wait()
                in_col = pc == pcp
                urg = ev_urg.get(k - 1)
                is_diag = in_col and pr == p % Pr
                is_head = in_col and p + 1 < nblk and pr == (p + 1) % Pr
                if (is_diag or is_head) and urg is not None:
                    urg.wait()                                         # block column p is up to date with panel k - 1
                if k >= 0:
                    hk = arr_H.get(k)
                    if is_diag or is_head:
                        wait_all([hk])
                    elif hk is not None:
                        self._late.append(hk)                          # (receive side of a broadcast this queue does not read)
                if is_diag:
                    blk = self._blk(p, p)
                    if k >= 0:
                        ops.gemm_nt(nb, nb, nb, -1.0, self.H[s].data_ptr(), nb, self.H[s].data_ptr(), nb, 1.0, blk.data_ptr(), ld, 1,
                                    q=CH)
                    ops.potrf_panel(nb, nb, blk.data_ptr(), ld, self.invd, self.info, p * nb)
                    ops.trinv(nb, blk.data_ptr(), ld, self.invd, self.W[s1].data_ptr(), nb)
                    w = min(nb, N - p * nb)
                    if w > 0:
                        ops.panel_scalars(blk, w, N - p * nb if p == last else -1, self.red)
                    self._mark(p, "W made")
                if in_col:
                    ev = None
                    if is_diag:
                        ev = ops.new_event()
                        ev.record()
                    arr_W[p] = _Arrival(self._xbcast("W", p, self.W[s1], (p % Pr, pcp), self.g_colw, Pr), ev)
                if p + 1 < nblk:
                    ev = None
                    if is_head:
                        hb = self._blk(p + 1, p)
                        if k >= 0:
                            wait_all([arr_R0.get(k)])
                            li_r0 = (p + 1) // Pr                      # R0(k) = L[p+1][k] sits at its block row in R[s]
                            ops.gemm_nt(nb, nb, nb, -1.0, _ptr(self.R[s], li_r0 * nb, 0), nb, self.H[s].data_ptr(), nb, 1.0,
                                        hb.data_ptr(), ld, 0, q=CH)
                        wait_all([arr_W[p]])
                        ops.gemm_nt(nb, nb, nb, 1.0, hb.data_ptr(), ld, self.W[s1].data_ptr(), nb, 0.0, self.H[s1].data_ptr(), nb, 0,
                                    q=CH)
                        if p + 1 == last:
                            ops.row_sumsq(self.H[s1][N - last * nb], self.red, q=CH)
                        self._mark(p, "H made")
                        ev = ops.new_event()
                        ev.record()
                    arr_H[p] = _Arrival(self._xbcast("H", p, self.H[s1], ((p + 1) % Pr, pcp), self.group, self.world), ev)
                ev_ch[p] = ops.new_event()
                ev_ch[p].record()

        def bulk(k):
            """Bulk queue, panel p = k + 1: the look-ahead update of block column p with panel k (rows p + 2 ..), the rank's rows of
            panel p -- the early block R0(p) = L[p+2][p] first, on its own communicator -- and the column exchange."""
            p = k + 1
            s, s1 = k % NBUF, p % NBUF
            pcp = p % Pc
            lp = p // Pc
            with ops.queue(BK):
                reuse(p)
                if k >= 0 and not self.lookahead:
                    ev_done[k].wait()
                in_col = pc == pcp
                li0 = self.li_ge(p + 2)                                # first block row of the slices of panel p
                rows0 = pr == (p + 2) % Pr and p + 2 < nblk            # this process row holds the early block R0(p)
                if in_col and k >= 0:
                    urg = ev_urg.get(k - 1)
                    if urg is not None:
                        urg.wait()
                    lu = self.li_ge(p + 1)                             # look-ahead update: rows I >= p + 1 ...
                    if pr == (p + 1) % Pr:
                        lu += 1                                        # ... except the head block (the chain queue's)
                    m = (self.nlr - lu) * nb
                    if m > 0:
                        wait_all([arr_H.get(k), arr_R0.get(k), arr_R.get(k)])
                        first = nb if (rows0 and lu == li0 and m > nb) else 0      # the block R0(p) comes from: first, alone
                        for (r0_, m_) in ((0, first), (first, m - first)):
                            if m_ > 0:
                                ops.gemm_nt(m_, nb, nb, -1.0, _ptr(self.R[s], lu * nb + r0_, 0), nb, self.H[s].data_ptr(), nb, 1.0,
                                            _ptr(A, lu * nb + r0_, lp * nb), ld, 0, q=BK)
                    self._mark(k, "LA done")
                # the rank's rows of panel p: R0 first where this process row has it, then the rest
                m = (self.nlr - li0) * nb
                if in_col and m > 0:
                    wait_all([arr_W[p]])
                n0 = nb if (rows0 and m > 0) else 0
                if n0:
                    ev = None
                    if in_col:
                        ops.gemm_nt(nb, nb, nb, 1.0, _ptr(A, li0 * nb, lp * nb), ld, self.W[s1].data_ptr(), nb, 0.0,
                                    _ptr(self.R[s1], li0 * nb, 0), nb, 0, q=BK)
                        if p + 2 == last:
                            ops.row_sumsq(self.R[s1][zrow_local], self.red_b, q=BK)
                        ev = ops.new_event()
                        ev.record()
                    arr_R0[p] = _Arrival(self._xbcast("R0", p, self.R[s1][li0 * nb:(li0 + 1) * nb], (pr, pcp), self.g_row0, Pc), ev)
                ev, works = None, []
                if m - n0 > 0:
                    if in_col:
                        ops.gemm_nt(m - n0, nb, nb, 1.0, _ptr(A, li0 * nb + n0, lp * nb), ld, self.W[s1].data_ptr(), nb, 0.0,
                                    _ptr(self.R[s1], li0 * nb + n0, 0), nb, 0, q=BK)
                        if pr == last % Pr and last >= p + 2 and not (n0 and p + 2 == last):
                            ops.row_sumsq(self.R[s1][zrow_local], self.red_b, q=BK)
                        ev = ops.new_event()
                        ev.record()
                        self._mark(p, "R made")
                    works = self._xbcast("R", p, self.R[s1][li0 * nb + n0:self.nlr * nb], (pr, pcp), self.g_row, Pc)
                arr_R[p] = _Arrival(works, ev)
                exchange(p)
                ev_bk[p] = ops.new_event()
                ev_bk[p].record()

        def exchange(p):
            """(bulk queue) The rank's columns J >= p + 2 of panel p, gathered from the process rows that hold them.  Source
            process row q holds the blocks J = J_q0 + t * lcm(P_r, P_c); in its R they are ``lcm / P_r`` block rows apart, in C
            ``lcm / P_c`` block columns."""
            s1 = p % NBUF
            lj0 = self.lj_ge(p + 2)
            if lj0 >= self.nlc:
                arr_C[p] = _Arrival([], None)
                return
            sc, sr = self.lcm // Pc, self.lcm // Pr
            bb = nb * nb
            wait_all([arr_R0.get(p), arr_R.get(p)])
            Cv = self.C[s1]
            qi = 0
            for q in range(Pr):
                J0 = next((J for J in self.my_cols[lj0:lj0 + sc] if J % Pr == q), None)
                if J0 is None:
                    continue
                nt = (nblk - 1 - J0) // self.lcm + 1
                ljq = J0 // Pc
                direct = sc == 1                       # one contributing process row: its share IS the rank's C
                dst = Cv[ljq * nb:(ljq + nt) * nb] if direct else self.piece[s1][qi][:nt * nb]
                qi += 1
                if q == pr:
                    liq = J0 // Pr
                    src = torch.as_strided(self.R[s1], (nt, bb), (sr * bb, 1), liq * bb)
                    ops.copy2d(dst.view(nt, bb), src, q=BK)
                for w_ in self._xbcast("C", p, dst, (q, pc), self.g_colx, Pr):
                    w_.wait()
                if not direct:
                    ops.copy2d(torch.as_strided(Cv, (nt, bb), (sc * bb, 1), ljq * bb), dst.view(nt, bb), q=BK)
            ev = ops.new_event()
            ev.record()
            self._mark(p, "C there")
            arr_C[p] = _Arrival([], ev)

        def step_main(k):
            """Panel k applied to the rank's blocks with I >= J >= k + 2 (main queue), the column the next look-ahead touches
            first."""
            s = k % NBUF
            with ops.queue("main"):
                wait_all([arr_R0.get(k), arr_R.get(k), arr_C.get(k)])
                self._mark(k, "arrived")
                li0 = self.li_ge(k + 2)
                lj0 = self.lj_ge(k + 2)
                m = (self.nlr - li0) * nb
                nseg = self.nlc - lj0
                if m > 0 and nseg > 0:
                    J0 = self.my_cols[lj0]
                    urgent = 1 if (self.lookahead and J0 == k + 2) else 0
                    for (q0, ns) in ((0, urgent), (urgent, nseg - urgent)):
                        if ns > 0:
                            ops.gemm_nt_gridstair(m, ns, nb, nb, -1.0, _ptr(self.R[s], li0 * nb, 0), nb,
                                                  _ptr(self.C[s], (lj0 + q0) * nb, 0), nb, J0 + q0 * Pc - pr, Pc, Pr, li0, 1.0,
                                                  _ptr(A, li0 * nb, (lj0 + q0) * nb), ld, q="main")
                        if q0 == 0 and urgent:
                            ev_urg[k] = ops.new_event()
                            ev_urg[k].record()
                ev_done[k] = ops.new_event()
                ev_done[k].record()
                self._mark(k, "applied")

        # ---- the schedule.  Look-ahead: the chain and the bulk of panel k + 1 are enqueued before update k and overlap it;
        # without it the same operations run one after the other.
        chain(-1)
        bulk(-1)
        for k in range(nblk):
            if self.lookahead:
                if k + 1 < nblk:
                    chain(k)
                    bulk(k)
                step_main(k)
            else:
                step_main(k)
                if k + 1 < nblk:
                    chain(k)
                    bulk(k)
            for d in (arr_W, arr_H, arr_R0, arr_R, arr_C, ev_urg, ev_done, ev_ch, ev_bk):
                d.pop(k - NBUF - 2, None)
        with ops.queue(BK):
            ev_bulk_end = ops.new_event()
            ev_bulk_end.record()
        # the reduction of _finish runs on the chain queue: behind the bulk queue's last accumulation and the receive sides of
        # the broadcasts nobody on this rank read
        with ops.queue(CH):
            for w_ in self._late:
                w_.wait()
            self._late = []
            ev_bulk_end.wait()

    def _mark(self, k, tag):
        if self._t0 is not None:
            e = self.ops.new_timing_event()
            e.record()
            self._marks.append((k, tag, e))

    def _finish(self, t_host0):
        ops, N = self.ops, self.N
        self.timings["host_enqueue_s"] = time.perf_counter() - t_host0
        with ops.queue("panel"):
            red = self.red
            red[:2] += self.red_b[:2]
            red[2] = self.info.to(torch.float64)[0]
            if self.world > 1 or self.force_collectives or self._model:
                info_t = red[2:3].clone()
                self._allreduce(red[:2], "sum")
                self._allreduce(info_t, "max")
                red[2] = info_t[0]
            if self._h_out is not None:
                self._h_out.copy_(red, non_blocking=True)
        ops.synchronize()
        logdet_half, zz, info = (float(v) for v in (self._h_out if self._h_out is not None else red))
        if self._t0 is not None:
            self.timings["steps_ms"] = [(k, tag, self._t0.elapsed_ms(e)) for k, tag, e in self._marks]
        if info != 0 and info <= N:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % int(info))
        ll_data = -0.5 * zz - logdet_half - 0.5 * N * math.log(2.0 * math.pi)
        if info != 0 or not math.isfinite(ll_data):
            raise np.linalg.LinAlgError("factorisation failed in the augmented row (non-finite y or K_tot?)")
        return ll_data, logdet_half
