"""Replicas: independent work items (hyperparameter starts, likelihood-grid points) spread over the ranks
of a ``torch.distributed`` job, one process per GPU.

The reference's only parallelism is a process pool over independent log-likelihood evaluations
(ref: gptools/gaussian_process.py:723-735 random starts of ``optimize_hyperparameters``; SURVEY.md
section 8f row 2).  A HIP context must not cross ``fork()``, so the pool is replaced by the ranks of the job
the user launched (``torchrun``): rank ``r`` of ``W`` evaluates items ``r, r+W, ...`` on its own GPU and the
results are all-gathered (pickled objects; any backend).  Below N of about 8-12 k this -- not a partitioned
factorisation -- is how several GPUs help a GP.
"""
__all__ = ["world_size", "shared", "distributed_map"]


def _dist():
    import torch.distributed as dist          # lazily: plain single-process use never pays for the import
    return dist


def world_size(group=None):
    import sys
    if "torch.distributed" not in sys.modules:    # nobody initialised a process group without importing it
        return 1
    dist = _dist()
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def shared(obj, src=0, group=None):
    """``obj`` as rank ``src`` has it, on every rank (random draws must agree before they are split)."""
    if world_size(group) == 1:
        return obj
    dist = _dist()
    box = [obj]
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    dist.broadcast_object_list(box, src=gsrc, group=group)
    return box[0]


def distributed_map(fn, items, group=None):
    """``[fn(x) for x in items]`` on every rank, each item evaluated by exactly one rank (round-robin).

    ``fn``'s results must be picklable.  An exception inside ``fn`` is re-raised on every rank after the
    gather, so that no rank is left waiting in a collective."""
    items = list(items)
    world = world_size(group)
    if world == 1:
        return [fn(x) for x in items]
    dist = _dist()
    rank = dist.get_rank(group)
    mine, err = [], None
    for i in range(rank, len(items), world):
        try:
            mine.append((i, fn(items[i])))
        except Exception as e:          # carried through the collective
            err = e
            break
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, repr(err) if err is not None else None), group=group)
    errors = [e for _, e in gathered if e is not None]
    if errors:
        raise RuntimeError("distributed_map: a rank failed: " + errors[0])
    out = [None] * len(items)
    for part, _ in gathered:
        for i, r in part:
            out[i] = r
    return out
