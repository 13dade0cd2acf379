"""Cell-sharded LOPQ search over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed search; north_star shards the index by LOPQ coarse cell.  Protocol:

* every rank holds the model, the cell-size table of the WHOLE index and the codes of the cells it
  owns (``LOPQSearcherHIP(model, shard=(rank, world, owner))``);
* every rank receives the whole query batch and derives the same multisequence order and quota cut
  (reference semantics, lopq/lopq/search.py:128-133) without talking to anyone, scans its own cells and
  ranks its candidates: ``search_partial_dev`` -> [nq, L] hits of 32 bytes;
* ONE collective per batch: all-gather of the per-rank hit lists (nq x L x 32 B per rank);
* every rank merges the ``world`` lists by (dist, visit_rank, pos) -- identical to the single-index
  result because a cell lives wholly on one rank.

torch.distributed is only the launcher/collective plumbing; backend "nccl" is RCCL on ROCm.  The same
functions run on CPU tensors with the "gloo" backend (tests/test_distributed_gloo.py).
"""
import numpy as np


def greedy_cell_owner(cell_counts, world):
    """Owner rank of every coarse cell: biggest cells first, each to the least loaded rank.
    Deterministic, so every rank computes the same table from the same counts."""
    counts = np.asarray(cell_counts, dtype=np.int64)
    owner = np.zeros(counts.shape[0], dtype=np.int32)
    load = np.zeros(world, dtype=np.int64)
    for cid in np.argsort(-counts, kind="stable"):
        r = int(np.argmin(load))
        owner[cid] = r
        load[r] += counts[cid]
    return owner


def all_gather_hits(hits, group=None):
    """All-gather the per-rank hit lists: [nq, L, 32] uint8 -> [world, nq, L, 32] (same device)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(hits.shape), dtype=hits.dtype, device=hits.device)
    try:
        dist.all_gather_into_tensor(out, hits.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):  # backends without the fused form
        parts = [torch.empty_like(hits) for _ in range(world)]
        dist.all_gather(parts, hits.contiguous(), group=group)
        out = torch.stack(parts)
    return out


class ShardedSearcher(object):
    """LOPQSearcherHIP sharded by coarse cell over the ranks of a torch.distributed group."""

    def __init__(self, model, owner=None, group=None):
        import torch.distributed as dist
        from .lopq.search import LOPQSearcherHIP
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.local = LOPQSearcherHIP(model, shard=(self.rank, self.world, owner))

    def add_codes_array(self, coarse, fine, ids=None, dedup=True):
        """Every rank is given ALL codes (it keeps its own cells and counts the rest)."""
        return self.local.add_codes_array(coarse, fine, ids, dedup)

    def get_nb_indexed(self):
        return self.local.get_nb_indexed()

    def search_batch_dev(self, q, quota=10, limit=None):
        from .lopq.search import merge_hits_dev
        hits, visited = self.local.search_partial_dev(q, quota=quota, limit=limit)
        out = merge_hits_dev(all_gather_hits(hits, self.group))
        out["visited"] = visited
        return out
