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
