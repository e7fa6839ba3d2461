"""Dealing independent work units (pocket graphs, batches of them) over the GPUs of a node.

SURVEY.md 8(e): every graph of a batch is an independent unit -- no reduction crosses graphs -- so the path shards with
no data-path collective.  What has to be balanced is the work per rank, which is proportional to the protein-protein
edge count (message chains are ~92 % of a step's FLOPs and pp edges ~95 % of the edges); real pockets differ 2-3x in
atom count, so dealing units round-robin by index leaves ranks idle.  ``shard_by_work`` is the classic
longest-processing-time greedy: units sorted by decreasing weight (ties by index), each given to the currently lightest
rank (ties by rank).  It is a pure function of (weights, world_size): every rank computes the same assignment without
communicating."""
from typing import List, Sequence


def shard_by_work(weights: Sequence[float], world_size: int) -> List[List[int]]:
    """-> one ascending index list per rank; the lists partition range(len(weights))."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    loads = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in sorted(range(len(weights)), key=lambda j: (-float(weights[j]), j)):
        r = min(range(world_size), key=lambda q: (loads[q], q))
        out[r].append(i)
        loads[r] += float(weights[i])
    for lst in out:
        lst.sort()
    return out


def shard_loads(weights: Sequence[float], shards: List[List[int]]) -> List[float]:
    return [sum(float(weights[i]) for i in lst) for lst in shards]


def host_share(local_rank: int, local_world: int, cpus: Sequence[int] = None):
    """-> (the CPUs of this rank's share, its thread count): the process's allowed CPUs cut into ``local_world`` contiguous
    slices (contiguous ids share caches and a NUMA node on the two-socket hosts of an 8-GPU node).  Pure function."""
    import os
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    local_world = max(1, int(local_world))
    per = max(1, len(cpus) // local_world)
    lo = (int(local_rank) % local_world) * per
    mine = cpus[lo:lo + per] or cpus[-per:]
    return mine, len(mine)


def pin_host_threads(local_rank: int, local_world: int, max_threads: int = 16) -> dict:
    """One process per GPU means ``local_world`` processes share the host: call this FIRST in a rank process, before anything
    touches the GPU or starts a thread pool.  It confines the process to its slice of the CPUs (sched_setaffinity) and sizes
    the OpenMP / torch intra-op pools to it -- a training step is as long on the host as on the device, and eight
    ranks that each start a 256-thread pool on a 256-thread host slow each other's launch loops down by integer factors.
    A single-rank run is left alone (its CPU legs want the whole host).  Returns what it did, for the bench line / logs."""
    import os
    cpus = sorted(os.sched_getaffinity(0))
    info = {"host_cpus": len(cpus), "local_world": int(local_world), "pinned": False, "host_threads_per_rank": len(cpus)}
    if int(local_world) <= 1:
        return info
    mine, n = host_share(local_rank, local_world, cpus)
    threads = max(1, min(n, int(max_threads)))
    try:
        os.sched_setaffinity(0, mine)
        info["pinned"] = True
        info["cpu_affinity"] = f"{mine[0]}-{mine[-1]}" if mine == list(range(mine[0], mine[-1] + 1)) else ",".join(map(str, mine))
    except OSError as e:                                 # (a container may forbid it: the thread counts still apply)
        info["affinity_error"] = str(e)
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = str(threads)
    try:
        import torch
        torch.set_num_threads(threads)
    except Exception as e:                               # noqa: BLE001  (torch absent / pool already fixed: recorded, not fatal)
        info["torch_threads_error"] = str(e)
    info["host_threads_per_rank"] = threads
    return info
