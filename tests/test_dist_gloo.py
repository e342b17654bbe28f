"""CPU suite, part 3: the N>1 path (gptools_amd/dist.py) under torch.distributed/gloo with world_size 2 and 3.

The partitioning, look-ahead ordering, panel broadcast and the scalar all-reduce are the product code;
the dense local operations are injected here as numpy/scipy stand-ins (test infrastructure: the product
ops class, HipPanelOps, refuses to run without a GPU)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _view(ptr, rows, cols, ld):
    base = np.ctypeslib.as_array((ctypes.c_double * (max(rows - 1, 0) * ld + cols)).from_address(ptr))
    return np.lib.stride_tricks.as_strided(base, shape=(rows, cols), strides=(ld * 8, 8))


def _numpy_ops():
    """Same interface as gptools_amd.dist.HipPanelOps, on CPU tensors, built from the oracle + scipy (the serial
    queue / event defaults of gptools_amd.dist.PanelOps: everything runs in program order)."""
    from gptools_amd.dist import PanelOps

    class NumpyPanelOps(PanelOps, _NumpyDense):
        pass
    return NumpyPanelOps()


class _NumpyDense(object):
    device = torch.device("cpu")

    def __init__(self):
        from oracle import oracle as O
        self.O = O

    def kbuild_block(self, kernel_id, params, X, n, r0, r1, c0, c1, err_y, noise_var, diag_add, out, ld):
        Xn, nn = X.numpy(), n.numpy()
        K = self.O.kbuild(kernel_id, params, Xn[r0:r1], nn[r0:r1], Xn[c0:c1], nn[c0:c1])
        e = err_y.numpy()
        for j in range(c0, c1):
            if r0 <= j < r1:
                K[j - r0, j - c0] = ((K[j - r0, j - c0] + noise_var) + e[j] ** 2) + diag_add
        _view(out, r1 - r0, c1 - c0, ld)[:, :] = K

    def kbuild_rect(self, kernel_id, params, Xi, ni, r0, r1, Xj, nj, c0, c1, out, ld):
        K = self.O.kbuild(kernel_id, params, Xi.numpy()[r0:r1], ni.numpy()[r0:r1], Xj.numpy()[c0:c1], nj.numpy()[c0:c1])
        _view(out, r1 - r0, c1 - c0, ld)[:, :] = K

    def potrf_panel(self, m, nb, A, lda, invd, info, info_base):
        import scipy.linalg
        P = _view(A, m, nb, lda)
        try:
            L = scipy.linalg.cholesky(np.tril(P[:nb]) + np.tril(P[:nb], -1).T, lower=True)
        except np.linalg.LinAlgError as e:
            if int(info[0]) == 0:
                info[0] = info_base + 1
            return
        P[:nb] = L
        if m > nb:
            P[nb:] = scipy.linalg.solve_triangular(L, P[nb:].T, lower=True).T

    def trsm_rlt(self, m, nb, L, ldl, invd, B, ldb, q="panel"):
        import scipy.linalg
        Lv, Bv = np.tril(_view(L, nb, nb, ldl)), _view(B, m, nb, ldb)
        Bv[:, :] = scipy.linalg.solve_triangular(Lv, Bv.T.copy(), lower=True).T

    def trinv(self, nb, L, ldl, invd, W, ldw, q="panel"):
        import scipy.linalg
        _view(W, nb, nb, ldw)[:, :] = scipy.linalg.solve_triangular(np.tril(_view(L, nb, nb, ldl)), np.eye(nb), lower=True)

    def gemm_nt(self, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, q="main"):
        Av, Bv, Cv = _view(A, m, k, lda).copy(), _view(B, n, k, ldb).copy(), _view(C, m, n, ldc)
        if tri:
            # like the library: only the lower trapezoid is written (what lies above the diagonal block's diagonal is never
            # read by anybody; poisoning it here proves that)
            res = alpha * Av.dot(Bv.T) + (beta * Cv if beta != 0.0 else 0.0)
            iu = np.triu_indices(n, 1)
            res[iu] = np.nan
            Cv[:, :] = res
            return
        Cv[:, :] = alpha * Av.dot(Bv.T) + (beta * Cv if beta != 0.0 else 0.0)


def _inputs(N, d, seed=4):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, d)
    n = np.zeros((N, d), dtype=np.int32)
    for i in range(3 * N // 4, N):
        n[i, i % d] = 1
    y = np.sin(X.sum(1)) + 0.05 * rs.randn(N)
    return X, n, y


def test_recorder_prunes_only_the_waits_its_vector_clocks_imply():
    """PlanRecorder's wait pruning (host side): a wait is dropped when the event was recorded on the waiting queue itself, when the
    queue has waited for it before, or when it is ordered behind the event through another event it waited for since; NOT when
    only the order of a communication channel would imply it (two exchanges on one channel may complete out of order under gloo:
    an "arrived" is a clock component of its own); a wait before the record is an error."""
    import torch
    from gptools_amd import dist as D
    rec = D.PlanRecorder(torch.device("cpu"))
    waits = lambda: [(o[1], o[2][0]) for o in rec.ops if o[0] == D.OP_WAIT and D.QUEUE_ID[o[1]] < 3]
    e1, e2, e3 = rec.new_event(), rec.new_event(), rec.new_event()
    with rec.queue("main"):
        e1.record()
        e1.wait()                                    # own queue: nothing
    assert waits() == []
    with rec.queue("panel"):
        e1.wait()
        e1.wait()                                    # again: nothing
        e2.record()
    assert waits() == [("panel", e1.idx)]
    with rec.queue("recv"):
        e2.wait()
        e1.wait()                                    # behind e2, which is behind e1: nothing
    assert waits() == [("panel", e1.idx), ("recv", e2.idx)]
    with rec.queue("main"):
        e3.record()
    with rec.queue("recv"):
        e3.wait()                                    # a later record of the main queue: needed
    assert waits()[-1] == ("recv", e3.idx) and rec.pruned == 3
    # two exchanges on one channel: waiting for the second does not stand for the first
    buf = torch.zeros(4, dtype=torch.float64)
    with rec.queue("panel"):
        (x1,) = rec.bcast(buf, 0, "comm1", ("R", (0, 0)))
        (x2,) = rec.bcast(buf, 0, "comm1", ("R", (0, 0)))
    with rec.queue("main"):
        x2.wait()
        n0 = len(waits())
        x1.wait()
        assert len(waits()) == n0 + 1
        x1.wait()                                    # ... but a second wait for the same arrival does
        assert len(waits()) == n0 + 1
    with rec.queue("recv"):
        fresh = rec.new_event()
        with pytest.raises(RuntimeError):
            fresh.wait()
    # the plan drops the records nobody waits for (e1 .. e3 are all waited for; "fresh" never was recorded)
    e4 = rec.new_event()
    with rec.queue("main"):
        e4.record()
    cp = D.CompiledPlan(rec)
    assert [o[2][0] for o in cp.ops if o[0] == D.OP_RECORD and o[1] == "main"] == [e1.idx, e3.idx]


def test_grid_recorded_op_list_is_static_and_replays_bit_identically():
    """The 2-D engine's compiled schedule, host side (one rank, numpy ops, no process group): two recordings give the same list,
    every wait follows its record, every record fits GPT_PLAN_W, the records of the grid-only ops (K rectangle, row_sumsq,
    gemm_nt_gridstair) carry the operands the interpreter hands to the ops object, and the replay reproduces the step loop's result
    bit for bit.  With the collectives switched on for a rank placed in a 2 x 4 grid (a model rank: nothing is sent) the list holds
    the five channels' broadcasts, each between a wait for "ready" and a record of "arrived" on its channel's queue."""
    import struct
    from gptools_amd import dist as D
    X, n, y = _inputs(900, 2)
    p = np.array([1.0, 0.3, 0.3])
    res = {}
    f64 = lambda bits: struct.unpack("<d", struct.pack("<q", bits))[0]
    for mode in ("python", False):
        plan = D.GridLML(X, n, (1, 1), nb=128, ops=_numpy_ops(), compiled=mode)
        res[mode] = (plan.fit(0, p, y, 0.05), plan.fit(0, 1.2 * p, y, 0.05))
        if not mode:
            continue
        (cp,) = plan._plans.values()
        rec = D.PlanRecorder(plan.device)
        real, plan.ops, plan._rec = plan.ops, rec, rec
        try:
            plan._schedule(None, None, 0.0, 0.0)
        finally:
            plan.ops, plan._rec = real, None
        assert [(o[0], o[1], o[2]) for o in cp.ops] == [(o[0], o[1], o[2]) for o in D.CompiledPlan(rec).ops]
        assert rec.pruned > 0 and len(cp.ops) < len(rec.ops)       # waits already implied / records nobody waits for: not in the list
        recorded, kinds = set(), set()
        for opcode, q, ints, py in cp.ops:
            assert len(ints) <= D.PLAN_W - 2 and q in D.QUEUE_ID
            kinds.add(opcode)
            if opcode == D.OP_RECORD:
                recorded.add(ints[0])
            elif opcode == D.OP_WAIT:
                assert ints[0] in recorded
            elif opcode == D.OP_GRIDSTAIR:
                assert [f64(v) if i in (4, 13) else v for i, v in enumerate(ints)] == list(py)
            elif opcode == D.OP_ROWSUMSQ:
                assert ints == [py[0].data_ptr(), py[0].numel(), py[1].data_ptr() + 8]
            elif opcode == D.OP_KRECT:
                Xi, ni, r0, r1, Xj, nj, c0, c1, out, ld = py
                assert ints == [Xi.data_ptr(), ni.data_ptr(), r0, r1, c0, c1, out, ld] and Xj is plan.X and nj is plan.n
        assert {D.OP_KBUILD, D.OP_PAD, D.OP_POTRF_PANEL, D.OP_TRINV, D.OP_GEMM, D.OP_SCALARS, D.OP_ROWSUMSQ, D.OP_GRIDSTAIR} <= kinds
    assert res["python"] == res[False]

    class Placed(D.GridLML):            # rank 5 of a 2 x 4 grid, collectives "on", nothing sent
        def _on(self, size):
            return size > 1

        def _allreduce(self, t, op):
            pass

    plan = Placed(X, n, (2, 4), nb=128, ops=_numpy_ops(), layout=5)
    rec = D.PlanRecorder(plan.device)
    real, plan.ops, plan._rec = plan.ops, rec, rec
    try:
        plan._schedule(None, None, 0.0, 0.0)
    finally:
        plan.ops, plan._rec = real, None
    chans = {}
    for i, (opcode, q, ints, py) in enumerate(rec.ops):
        if opcode == D.OP_BCAST:
            assert D.QUEUE_ID[q] >= 3 and rec.ops[i - 1][0] == D.OP_WAIT and rec.ops[i - 1][1] == q
            assert rec.ops[i + 1][0] == D.OP_RECORD and rec.ops[i + 1][1] == q and rec.ops[i + 1][2] == [py[3]]
            kind, src = py[1]
            assert q == D.GridLML.CHANNEL[kind] and ints[1] == py[0].numel()
            size = {"H": 8, "R": 4, "R0": 4, "W": 2, "C": 2}[kind]
            assert 0 <= ints[2] < size
            chans.setdefault(kind, 0)
            chans[kind] += 1
    assert set(chans) == {"H", "R", "R0", "W", "C"}, chans


def _worker(rank, world, port, N, d, nb, kernel_id, lookahead, bad, q, plan_kw=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gptools_amd.dist import DistributedLML, GridLML
        X, n, y = _inputs(N, d)
        if bad:
            X[1] = X[0]
            n[:] = 0
        plan_kw = dict(plan_kw or {})
        if "grid" in plan_kw:
            plan = GridLML(X, n, plan_kw.pop("grid"), nb=nb, ops=_numpy_ops(), lookahead=lookahead, **plan_kw)
        else:
            plan = DistributedLML(X, n, nb=nb, ops=_numpy_ops(), lookahead=lookahead, **plan_kw)
        p = np.concatenate(([1.0], 0.3 * np.ones(d)))
        try:
            res = plan.fit(kernel_id, p, y, 0.0 if bad else 0.05, diag_factor=0.0 if bad else 1e2)
        except np.linalg.LinAlgError as e:
            res = ("LinAlgError", str(e))
        # second evaluation with other hyperparameters reuses the buffers (MAP-loop usage)
        res2 = None if bad else plan.fit(kernel_id, 1.1 * p, y, 0.05)
        q.put((rank, res, res2, plan.my_blocks))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(world, N, d, nb, kernel_id, lookahead, bad=False, plan_kw=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, d, nb, kernel_id, lookahead, bad, q, plan_kw))
             for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


_WHOLE = {"schedule": "bcast"}
_WHOLE_INV = {"schedule": "bcast", "inv_min_rows": 0}        # rows below the diagonal block by inverse + GEMM
_WHOLE_SAG = {"schedule": "bcast", "exchange": "scatter_gather", "sag_min_bytes": 0}
# (default on CPU ops: the step loop is recorded once as an op list and every evaluation replays it through the Python interpreter of
# that list -- the list gpt_plan_run replays natively on a GPU; compiled=False: the step loop issues every operation itself)
_WHOLE_DIRECT = {"schedule": "bcast", "compiled": False}
_WHOLE_SAG_DIRECT = {"schedule": "bcast", "exchange": "scatter_gather", "sag_min_bytes": 0, "compiled": False}


@pytest.mark.parametrize("world,N,d,nb,kid,lookahead,plan_kw", [
    (2, 700, 3, 128, 1, True, _WHOLE),      # Matern52 with derivative rows, 6 block columns over 2 ranks
    (2, 700, 3, 128, 1, False, _WHOLE),
    (3, 500, 2, 128, 0, True, _WHOLE),      # SE, uneven block ownership (5 block columns over 3 ranks)
    (2, 100, 2, 256, 0, True, _WHOLE),      # fewer block columns than ranks -> an idle rank must still take part
    (2, 700, 3, 128, 1, True, _WHOLE_SAG),  # panels moved by scatter + all-gather
    (3, 700, 3, 128, 1, True, _WHOLE_INV),
    (2, 700, 3, 128, 1, False, _WHOLE_INV),
    (8, 1900, 2, 128, 0, True, _WHOLE_INV),
    (3, 500, 2, 128, 0, True, _WHOLE_DIRECT),
    (2, 700, 3, 128, 1, True, _WHOLE_SAG_DIRECT),
])
def test_distributed_fit_matches_single_process_oracle(world, N, d, nb, kid, lookahead, plan_kw):
    from oracle import oracle as O
    out = _run(world, N, d, nb, kid, lookahead, plan_kw=plan_kw)
    X, n, y = _inputs(N, d)
    p = np.concatenate(([1.0], 0.3 * np.ones(d)))
    ref = O.fit(kid, p, X, n, y, 0.05 * np.ones(N))
    ref2 = O.fit(kid, 1.1 * p, X, n, y, 0.05 * np.ones(N))
    owned = []
    for rank, res, res2, blocks in out:
        assert abs(res[0] - ref["ll_data"]) <= 1e-9 * abs(ref["ll_data"]), (rank, res, ref["ll_data"])
        assert abs(res[1] - ref["logdet_half"]) <= 1e-10 * abs(ref["logdet_half"])
        assert abs(res2[0] - ref2["ll_data"]) <= 1e-9 * abs(ref2["ll_data"])
        owned += blocks
    assert sorted(owned) == list(range((N + 1 + nb - 1) // nb))     # block-cyclic cover, no overlap
    assert len({r[1] for r in out}) == 1                            # every rank reports the same numbers


# (the grid engine records its step loop as an op list too -- default here: the Python interpreter of the list, the one
# gpt_plan_run replays natively on a GPU with five RCCL communicators per rank; a trailing False: the step loop itself)
@pytest.mark.parametrize("world,grid,N,d,nb,kid,lookahead", [
    (4, (2, 2), 1100, 3, 128, 1, True),      # 9 block rows / columns over 2 x 2, Matern52 with derivative rows
    (4, (2, 2), 1100, 3, 128, 1, False),
    (8, (2, 4), 1900, 2, 128, 0, True),      # the grids of the 8-GPU node: P_r | P_c (one contributing process row per column)
    (8, (2, 4), 1900, 2, 128, 0, False),
    (8, (4, 2), 1900, 2, 128, 0, True),      # P_c | P_r: two process rows contribute to every rank's columns
    (8, (4, 2), 1900, 2, 128, 0, False),
    (6, (2, 3), 1500, 2, 128, 1, True),      # coprime: every process row contributes, uneven ownership
    (4, (2, 2), 100, 2, 256, 0, True),       # one block: three idle ranks must still take part
    (2, (1, 2), 700, 3, 128, 1, True),       # the 1-D block-column layout as a grid
    (2, (2, 1), 700, 3, 128, 1, True),       # a block-row layout: the column exchange is an all-gather over all ranks
    (8, (2, 4), 1023, 2, 128, 0, True),      # N + 1 a multiple of nb: no padding rows beyond the augmented row
    (4, (2, 2), 1024, 2, 128, 0, True),      # N a multiple of nb: the augmented row opens a block row of its own
    (4, (4, 1), 1900, 2, 128, 0, True),      # nlc % (lcm / P_c) != 0 -- the exchange's staging piece in whole blocks
    (12, (4, 3), 2000, 2, 128, 0, True),     # the same with three process columns
    (8, (4, 2), 1900, 2, 128, 0, (True, False)),      # the step loop issuing every operation itself (compiled=False)
    (6, (2, 3), 1500, 2, 128, 1, (False, False)),
])
def test_grid_fit_matches_single_process_oracle(world, grid, N, d, nb, kid, lookahead):
    """2-D block-cyclic layout (gptools_amd.dist.GridLML): every rank returns the oracle's ll / log-determinant
    (1e-9 / 1e-10), all ranks the same bits, the blocks of the lower triangle are covered exactly once, and a second
    evaluation with other hyperparameters reuses the buffers."""
    from oracle import oracle as O
    plan_kw = {"grid": grid}
    if isinstance(lookahead, tuple):
        lookahead, plan_kw["compiled"] = lookahead
    out = _run(world, N, d, nb, kid, lookahead, plan_kw=plan_kw)
    X, n, y = _inputs(N, d)
    p = np.concatenate(([1.0], 0.3 * np.ones(d)))
    ref = O.fit(kid, p, X, n, y, 0.05 * np.ones(N))
    ref2 = O.fit(kid, 1.1 * p, X, n, y, 0.05 * np.ones(N))
    owned = []
    for rank, res, res2, blocks in out:
        assert abs(res[0] - ref["ll_data"]) <= 1e-9 * abs(ref["ll_data"]), (rank, res, ref["ll_data"])
        assert abs(res[1] - ref["logdet_half"]) <= 1e-10 * abs(ref["logdet_half"])
        assert abs(res2[0] - ref2["ll_data"]) <= 1e-9 * abs(ref2["ll_data"])
        owned += blocks
    nblk = (N + 1 + nb - 1) // nb
    assert sorted(owned) == sorted((I, J) for J in range(nblk) for I in range(J, nblk))
    assert len({r[1] for r in out}) == 1 and len({r[2] for r in out}) == 1


def test_grid_not_positive_definite_raises_on_every_rank():
    for grid, world in (((2, 2), 4), ((2, 4), 8)):
        out = _run(world, 300, 2, 128, 0, True, bad=True, plan_kw={"grid": grid})
        for rank, res, _, _ in out:
            assert res[0] == "LinAlgError" and "not positive definite" in res[1]


def test_distributed_not_positive_definite_raises_on_every_rank():
    for plan_kw in (_WHOLE, _WHOLE_INV):
        out = _run(2, 300, 2, 128, 0, True, bad=True, plan_kw=plan_kw)
        for rank, res, _, _ in out:
            assert res[0] == "LinAlgError" and "not positive definite" in res[1]


# ---- replicas: independent items over the ranks (gptools_amd/replicas.py) ------------------------------------
def _replica_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gptools_amd import replicas
        draws = replicas.shared(np.random.RandomState(100 + rank).rand(7, 2))    # rank 0's draws everywhere
        seen = []

        def f(x):
            seen.append(float(x[0]))
            return {"rank": rank, "val": float(x.sum()) ** 2}
        out = replicas.distributed_map(f, list(draws))
        try:
            replicas.distributed_map(lambda x: 1 // (0 if x == 4 else 1), range(6))    # item 4 belongs to rank 1
            err = None
        except RuntimeError as e:
            err = str(e)
        q.put((rank, draws, out, len(seen), err))
    finally:
        dist.destroy_process_group()


def test_replicas_map_over_ranks():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_draws = np.random.RandomState(100).rand(7, 2)
    for rank, draws, res, nseen, err in out:
        np.testing.assert_array_equal(draws, ref_draws)                 # everyone works on rank 0's draws
        assert [r["rank"] for r in res] == [i % world for i in range(7)]  # round-robin ownership
        np.testing.assert_allclose([r["val"] for r in res], ref_draws.sum(1) ** 2)
        assert nseen == len(range(rank, 7, world))                      # each item evaluated exactly once
        assert err is not None and "ZeroDivisionError" in err           # a failure surfaces on every rank


def test_replicas_single_process_is_plain_map():
    from gptools_amd import replicas
    assert replicas.world_size() == 1
    assert replicas.shared({"a": 1}) == {"a": 1}
    assert replicas.distributed_map(lambda v: v * v, [1, 2, 3]) == [1, 4, 9]


def test_recorded_op_list_is_static_and_encodes_what_the_interpreter_runs():
    """Compiled schedules, host side (no GPU, no process group: one rank, numpy ops): the recorder's integer records -- what
    gpt_plan_run replays -- carry the same operands as the Python arguments the interpreter hands to the ops object (addresses,
    sizes, the bit patterns of alpha / beta), events are recorded before they are waited for, every record fits GPT_PLAN_W, a second
    recording yields the same list, and replaying the list reproduces the step loop's result bit for bit."""
    import struct
    from gptools_amd import dist as D
    X, n, y = _inputs(900, 2)
    p = np.array([1.0, 0.3, 0.3])
    res = {}
    for mode in ("python", False):
        plan = D.DistributedLML(X, n, nb=128, ops=_numpy_ops(), compiled=mode, inv_min_rows=256)
        res[mode] = (plan.fit(0, p, y, 0.05), plan.fit(0, 1.2 * p, y, 0.05))
        if mode:
            (cp,) = plan._plans.values()
            lists = [cp.ops]
            rec = D.PlanRecorder(plan.device)
            real, plan.ops, plan._rec = plan.ops, rec, rec
            try:
                plan._schedule(None, None, 0.0, 0.0)
            finally:
                plan.ops, plan._rec = real, None
            lists.append(D.CompiledPlan(rec).ops)     # (the plan drops the records of events nobody waits for)
            assert [(o[0], o[1], o[2]) for o in lists[0]] == [(o[0], o[1], o[2]) for o in lists[1]]
            recorded = set()
            f64 = lambda bits: struct.unpack("<d", struct.pack("<q", bits))[0]
            kinds = set()
            for opcode, q, ints, py in cp.ops:
                assert len(ints) <= D.PLAN_W - 2 and q in D.QUEUE_ID
                kinds.add(opcode)
                if opcode == D.OP_RECORD:
                    recorded.add(ints[0])
                elif opcode == D.OP_WAIT:
                    assert ints[0] in recorded
                elif opcode == D.OP_GEMM:
                    m, n_, k, alpha, A, lda, B, ldb, beta, C, ldc, tri = py
                    assert ints == [m, n_, k, ints[3], A, lda, B, ldb, ints[8], C, ldc, tri]
                    assert f64(ints[3]) == alpha and f64(ints[8]) == beta
                elif opcode == D.OP_STAIR:
                    assert f64(ints[4]) == py[4] and f64(ints[11]) == py[11] and ints[5] == py[5] and ints[12] == py[12]
                elif opcode == D.OP_COPY2D:
                    dst, src = py
                    assert ints == [src.shape[0], src.shape[1], src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0)]
                elif opcode == D.OP_POTRF_PANEL:
                    assert ints[:4] == list(py[:4]) and ints[4] == py[4].data_ptr() and ints[5] == py[5].data_ptr() and ints[6] == py[6]
                elif opcode == D.OP_KBUILD:
                    assert ints == list(py)
                elif opcode == D.OP_SCALARS:
                    assert ints == [py[0].data_ptr(), py[0].stride(0), py[1], py[2], py[3].data_ptr()]
            assert {D.OP_KBUILD, D.OP_PAD, D.OP_COPY2D, D.OP_POTRF_PANEL, D.OP_TRINV, D.OP_GEMM, D.OP_STAIR, D.OP_SCALARS} <= kinds
    assert res["python"] == res[False]
