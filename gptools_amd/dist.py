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
  * Two schedules (``schedule=``): ``"bcast"`` factors a panel whole and then broadcasts it; ``"pipelined"``
    cuts every panel into a few row chunks of growing size (2, 1, 5, 24, ... blocks): the owner factors the
    diagonal block with the first chunk and sends it at once, the TRSM of the later chunks, their transfer and
    the next owner's column update proceed chunk by chunk behind it, so the serial chain through the panels
    carries only the small head chunks (its own communicator) instead of whole panels.  Two ways to move a
    chunk (``exchange=``): one broadcast, or scatter + all-gather (every xGMI link of the root carries 1/world
    of the chunk, then all links carry the all-gather) for chunks above ``sag_min_bytes``.

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

    def pad_block(self, A, lj, c0, nb, N, NP, y, big):
        """Rows [N, NP) of the local block column ``lj`` (global first column ``c0``): augmented row y^T, unit diagonal on
        the padding, ``big`` under the augmented row, zeros elsewhere."""
        blk = A[N:, lj * nb:(lj + 1) * nb]
        blk.zero_()
        c1 = min(c0 + nb, N)
        if c0 < N:
            blk[0, :c1 - c0] = y[c0:c1]
        p0 = max(c0, N)
        if p0 < c0 + nb:
            idx = torch.arange(p0, c0 + nb, device=A.device)
            A[idx, lj * nb + (idx - c0)] = 1.0
            if c0 <= N < c0 + nb:
                A[N, lj * nb + (N - c0)] = big

    def panel_scalars(self, buf, w, zrow, red, q="panel"):
        """red[0] += sum(log diag(buf[:w, :w])); red[1] += |buf[zrow, :w]|^2."""
        red[0] += torch.log(torch.diagonal(buf[:w, :w])).sum()
        z = buf[zrow, :w]
        red[1] += (z * z).sum()

    def gemm_nt_stair(self, m, nseg, seg_cols, k, alpha, A, lda, B, ldb, b_stride, row_step, beta, C, ldc, q="main"):
        """Default: one lower-trapezoid ``gemm_nt`` per column segment (8-byte elements)."""
        for s in range(nseg):
            r = s * row_step
            self.gemm_nt(m - r, seg_cols, k, alpha, A + r * lda * 8, lda, B + s * b_stride * ldb * 8, ldb, beta,
                         C + (r * ldc + s * seg_cols) * 8, ldc, 1, q=q)


class _StreamEvent(object):
    def __init__(self, timing=False):
        self.ev = torch.cuda.Event(enable_timing=timing)

    def elapsed_ms(self, later):
        return self.ev.elapsed_time(later.ev)

    def record(self):
        self.ev.record(torch.cuda.current_stream())

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


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
        self._ctx = {"main": self.ctx_main, "panel": self.ctx_panel}
        self._stream = {"main": self.main_stream, "panel": self.panel_stream, "recv": self.recv_stream}
        for c in self._ctx.values():
            c.set_option("lookahead", 0)
        # everything the panel context launches sits on the chain and shares CUs with the main context's trailing update:
        # its GEMM main loops keep a raised wave priority (gemm.hip; the TRSM / fused kernels carry theirs themselves)
        self.ctx_panel.set_option("gemm_prio", int(os.environ.get("GPT_DIST_PANEL_PRIO", "2")))

    def queue(self, q):
        return torch.cuda.stream(self._stream[q])

    def new_event(self):
        return _StreamEvent()

    def new_timing_event(self):
        return _StreamEvent(timing=True)

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
        _lib.check(self.lib.gpt_dev_copy2d(self._ctx[q].handle, src.shape[0], src.shape[1], src.data_ptr(), src.stride(0),
                                           dst.data_ptr(), dst.stride(0)))

    def pad_block(self, A, lj, c0, nb, N, NP, y, big):
        _lib.check(self.lib.gpt_dev_pad_block(self.ctx_main.handle, _ptr(A, 0, lj * nb), A.stride(0), c0, nb, N, NP,
                                              y.data_ptr(), float(big)))

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

    ``schedule``: ``"bcast"`` (whole panels, the default) or ``"pipelined"`` (row-chunked panels; needs look-ahead);
    ``exchange``: ``"bcast"`` or ``"scatter_gather"`` (chunks of at least ``sag_min_bytes`` whose row count divides by
    the world size; smaller ones are broadcast); ``chunk_blocks``: panel-local block rows at which a panel is cut.
    All three may be changed between ``fit`` calls (bench.py times the combinations during warm-up).  Several plans
    (e.g. of different ``nb``) may share one ``ops`` object and one ``group_tail`` communicator.
    """
    NBUF = 4

    def __init__(self, X, n, nb=512, group=None, ops=None, device=None, lookahead=True, layout=None,
                 schedule="bcast", exchange="bcast", chunk_blocks=(2, 3, 8, 32), sag_min_bytes=8 << 20,
                 owner_first=None, inv_trsm=True, inv_min_rows=8192, group_tail=None):
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
        if schedule not in ("pipelined", "bcast") or exchange not in ("bcast", "scatter_gather"):
            raise ValueError("schedule must be 'pipelined' or 'bcast', exchange 'bcast' or 'scatter_gather'")
        self.schedule = schedule
        self.exchange = exchange
        self.chunk_blocks = tuple(int(b) for b in chunk_blocks)
        if not self.chunk_blocks or self.chunk_blocks[0] < 2 or list(self.chunk_blocks) != sorted(set(self.chunk_blocks)):
            raise ValueError("chunk_blocks must be increasing and start at 2 or more (the head chunk holds L_kk and L_k+1,k)")
        self.sag_min_bytes = int(sag_min_bytes)
        # owner_first: the owner of panel k+1 starts its own trailing update of step k only after it has produced panel
        # k+1.  Everybody else is waiting for that panel, nobody for the owner's update; and a panel kernel that has to
        # share the chip with a trailing update already in flight gets a third of it, whatever the stream priority
        # (N=32768, 8 ranks replayed on one GPU: 3.0 ms from first to last chunk of an early panel, against
        # ~0.8 ms when it has the chip).  Pointless with one or two ranks (the owner is always the same / every
        # other step), on by default from three.
        self.owner_first = (self.world > 2) if owner_first is None else (owner_first if owner_first == "head" else bool(owner_first))
        # inv_trsm: the rows of a panel below its head chunk (pipelined) / diagonal block (bcast) are solved as ONE GEMM against the
        # explicit inverse of the factored diagonal block (computed once per panel, off the chain) instead of by
        # substitution in four 128-column leaves: twice the flops at several times the rate for tall chunks.
        self.inv_trsm = bool(inv_trsm)
        self.inv_min_rows = int(inv_min_rows)       # whole-panel schedule: panels of at least this many rows take that route
        self._inv_panel = False
        # the later chunks of the pipelined schedule travel on a communicator of their own, so that a head chunk never
        # queues behind the bulk of an earlier panel (collectives of one communicator run in issue order)
        self.group_tail = group if group_tail is None else group_tail
        # RCCL runs the collectives of a communicator in issue order on its stream; other backends need explicit waits
        self._stream_ordered = dist.is_initialized() and dist.get_backend(group) == "nccl"
        if group_tail is None and dist.is_initialized() and layout is None and (self.world > 1 or self.force_collectives):
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            self.group_tail = dist.new_group(ranks=ranks)
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

    def _exchange(self, buf, src, group=None, tag=None):
        """Start moving the contiguous rows ``buf`` from rank ``src`` to everyone; returns the list of work handles to
        wait for (empty when no collective is needed).  scatter + all-gather: the root's links each carry 1/world of
        the rows, then every link carries the all-gather; a broadcast moves the whole chunk along one path.  ``tag`` = (panel, first row) is not used here
        (scratch/sim_ranks.py overrides this method and needs to know what is being moved)."""
        if not self._collectives_on():
            return []
        W = self.world
        rows = buf.shape[0]
        if (self.exchange == "scatter_gather" and rows % W == 0
                and buf.numel() * buf.element_size() >= self.sag_min_bytes):
            group = self.group if group is None else group
            gsrc = dist.get_global_rank(group, src) if group is not None else src
            c = rows // W
            pieces = [buf[r * c:(r + 1) * c] for r in range(W)]
            mine = pieces[self.grank]
            w1 = dist.scatter(mine, scatter_list=pieces if self.grank == src else None, src=gsrc, group=group,
                              async_op=True)
            if not self._stream_ordered:
                w1.wait()      # (gloo runs asynchronous operations on a thread pool, in no particular order)
            w2 = dist.all_gather_into_tensor(buf, mine, group=group, async_op=True)      # in place
            return [w1, w2]
        return [self._bcast(buf, src, async_op=True, group=group)]

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

    def _begin(self, kernel_id, params, y, err_y, noise_var, diag_factor):
        """Common head of both schedules: upload y / err_y, build the local block columns (main queue)."""
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
            self._t0 = ops.new_timing_event() if self.trace else None
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
        if kernel_id in (_lib.KERNEL_RQ, _lib.KERNEL_MATERN) and 2 * self._n_maxsum > 8:
            raise ValueError("RationalQuadratic / Matern kernel: derivative orders of a pair sum to %d, the device builder "
                             "supports 8" % (2 * self._n_maxsum))
        if self.schedule == "pipelined" and self.lookahead:
            return self._fit_pipelined(kernel_id, params, y, err_y, noise_var, diag_factor)
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
        ops = self.ops
        t_host0 = time.perf_counter()
        nb, NP, world, rank, NBUF = self.nb, self.NP, self.world, self.rank, self.NBUF
        nblk = self.nblk
        owner = lambda J: J % world == rank
        ev_asm = self._begin(kernel_id, params, y, err_y, noise_var, diag_factor)
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
        return self._finish(t_host0)


    # ------------------------------------------------------------------------------------------
    def _chunk_bounds(self, k):
        """Panel-local block rows at which panel k is cut: [0, 2, 8, 32, ..., blocks of panel k] (``chunk_blocks``);
        a cut that would leave a last chunk smaller than half of what precedes it is dropped."""
        mb = self.nblk - k
        b = [0] + [e for e in self.chunk_blocks if e < mb and 2 * (mb - e) >= e]
        return b + [mb]

    def _fit_pipelined(self, kernel_id, params, y, err_y, noise_var, diag_factor):
        """Row-chunked schedule.  Panel k lives in P[k % NBUF], cut at ``_chunk_bounds(k)``.  Its owner, on the panel
        queue, takes the chunks top down: rows [lo, hi) of the staged column get panel k-1 applied as soon as rows
        [lo + nb, hi + nb) of panel k-1 are there (block row 1 of panel k-1 is the other operand), then the first chunk
        is factored (diagonal block + the TRSM of its remaining rows), later ones are solved against the factored
        diagonal block, and each chunk starts travelling at once -- head chunks on ``group``, the rest on
        ``group_tail``.  The chain from panel to panel therefore is: head of k-1 arrives -> 2-block update ->
        diagonal block -> head of k leaves, while the bulk TRSM, the bulk transfer and the next owner's column update
        overlap chunk by chunk.  (Precisely: the head of panel k reads local blocks 1 AND 2 of panel k-1; block 2 is cut
        off as a one-block chunk of its own -- ``chunk_blocks`` = (2, 3, 8, 32) -- and travels on the chain communicator
        right behind the head, so the chain is head -> block 2 -> next head and never waits for a bulk chunk on the tail
        communicator.  With cuts that leave block 2 inside a larger chunk the results are the same and the chain is
        longer.)  Other ranks post their side of every exchange on the idle "recv" queue.
          main queue, step k: every chunk of panel k has arrived -> apply it to the owned block columns right of k+1
                              (k+1 is its owner's business on the panel queue), column k+2 first ("urgent").
        Buffer reuse: P[k % NBUF] is written again (staged or received into) only after this rank's main queue is done
        with step k - NBUF and its panel queue has finished reading panel k - NBUF."""
        ops = self.ops
        t_host0 = time.perf_counter()
        nb, NP, world, rank, NBUF = self.nb, self.NP, self.world, self.rank, self.NBUF
        nblk = self.nblk
        owner = lambda J: J % world == rank
        ev_asm = self._begin(kernel_id, params, y, err_y, noise_var, diag_factor)
        for q in ("panel", "recv"):
            # nothing on the other queues may run ahead of this rank's own K build / zeroed scalars (a rank that does
            # not own panel 0 would otherwise stage its first block column while the builder is still writing it)
            with ops.queue(q):
                ev_asm.wait()
        ev_urg, ev_done, ev_pq = {}, {}, {}
        arrivals = {}

        def produce(k):
            buf = self.P[k % NBUF]
            own = owner(k)
            bl = self._chunk_bounds(k)
            prev = arrivals.get(k - 1)
            pbuf = self.P[(k - 1) % NBUF]
            pb = self._chunk_bounds(k - 1) if k > 0 else None
            arr = arrivals[k] = []
            with ops.queue("panel" if own else "recv"):
                if k - NBUF in ev_done:
                    ev_done.pop(k - NBUF).wait()
                if k - NBUF in ev_pq:
                    ev_pq.pop(k - NBUF).wait()
                inv = own and self.inv_trsm and len(bl) > 2
                if own:
                    if k - 2 in ev_urg:
                        ev_urg.pop(k - 2).wait()           # column k is up to date with panel k-2
                    if inv:
                        # head rows into the panel buffer, the rest into the scratch column S: those rows reach the
                        # panel buffer as S[rows] * L_kk^-T
                        h = bl[1] * nb
                        lk = k // world
                        col = self.A[k * nb:, lk * nb:(lk + 1) * nb]
                        ops.copy2d(buf[:h], col[:h])
                        ops.copy2d(self.S[h:NP - k * nb], col[h:])
                    else:
                        self._stage_panel(k, buf, allow_inv=False)      # before any wait for panel k-1
                src = self.S if inv else buf
                waited = 0
                for c in range(len(bl) - 1):
                    lo, hi = bl[c] * nb, bl[c + 1] * nb
                    ev = None
                    if own:
                        if k > 0:
                            # last block row of panel k-1 this chunk reads: local block bl[c+1]
                            need = max(j for j in range(len(pb) - 1) if pb[j] <= bl[c + 1])
                            while waited <= need:
                                prev[waited].wait()
                                waited += 1
                            ops.gemm_nt(hi - lo, nb, nb, -1.0, _ptr(pbuf, lo + nb, 0), nb, _ptr(pbuf, nb, 0), nb, 1.0,
                                        _ptr(buf if c == 0 else src, lo, 0), nb, 1 if c == 0 else 0, q="panel")
                        if c == 0:
                            ops.potrf_panel(hi, nb, buf.data_ptr(), nb, self.invd, self.info, k * nb)
                        elif inv:
                            ops.gemm_nt(hi - lo, nb, nb, 1.0, _ptr(src, lo, 0), nb, self.Winv.data_ptr(), nb, 0.0,
                                        _ptr(buf, lo, 0), nb, 0, q="panel")
                        else:
                            ops.trsm_rlt(hi - lo, nb, buf.data_ptr(), nb, self.invd, _ptr(buf, lo, 0), nb)
                        ev = ops.new_event()
                        ev.record()
                    # The chain communicator carries the head AND local block 2: the head of panel k+1 (local blocks
                    # 0-1 = global k+1, k+2) is updated with local blocks 1 and 2 of panel k, so block 2 must not queue
                    # behind this panel's bulk (with the default cuts it is a chunk of its own, sent right after the head).
                    works = self._exchange(buf[lo:hi], k % world, group=self.group if bl[c + 1] <= 3 else self.group_tail,
                                           tag=(k, lo))
                    arr.append(_Arrival(works, ev))
                    if inv and c == 0:
                        # after the head is on its way: the inverse the later chunks are multiplied by
                        ops.trinv(nb, buf.data_ptr(), nb, self.invd, self.Winv.data_ptr(), nb)
                if own:
                    self._factored.append((k, buf))
                    self._accumulate_scalars()
                    if k > 0:
                        ev_pq[k - 1] = ops.new_event()     # this queue no longer reads panel k-1 ...
                        ev_pq[k - 1].record()
                    ev_pq[k] = ops.new_event()             # ... nor (scalars) its own panel
                    ev_pq[k].record()

        produce(0)
        for k in range(nblk):
            buf = self.P[k % NBUF]
            if k + 1 < nblk:
                produce(k + 1)
            with ops.queue("main"):
                if self.owner_first and k + 1 < nblk and owner(k + 1):
                    # own panel first (see __init__): all of it, or ("head") only the chunks on the chain -- the head and
                    # block 2, which the next owner's head needs; the bulk below them is then produced beside this rank's
                    # own trailing update (it is needed by the main queues, which lag the chain while updates dominate)
                    own = arrivals[k + 1]
                    own[min(1, len(own) - 1) if self.owner_first == "head" else -1].ev.wait()
                for a in arrivals.pop(k):
                    a.wait()
                self._mark(k, "arrived")
                mine = [J for J in self.my_blocks if J > k + 1]
                urgent = k + 2
                if urgent in mine:
                    self._update_block(k, urgent, buf)
                    mine.remove(urgent)
                    ev_urg[k] = ops.new_event()
                    ev_urg[k].record()
                if mine:
                    self._update_blocks(k, mine, buf)
                ev_done[k] = ops.new_event()
                ev_done[k].record()
                self._mark(k, "applied")
        return self._finish(t_host0)
