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


def all_gather_stack(x, group=None):
    """All-gather equal-shaped tensors -> [world, *x.shape].  The collective writes into the CONCATENATED form
    ([world * n, ...]), which both RCCL and gloo accept; the stacked view is free."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    x = x.contiguous()
    shape, dtype = tuple(x.shape), x.dtype
    if dtype not in (torch.uint8, torch.int32, torch.int64, torch.float32, torch.float64):
        # RCCL / gloo know no 16-bit integers (the coarse codes): the bytes travel as uint8
        x = x.reshape(-1).view(torch.uint8)
    flat = x.reshape(-1)
    out = torch.empty(world * flat.shape[0], dtype=flat.dtype, device=flat.device)
    try:
        dist.all_gather_into_tensor(out, flat, group=group)
    except (RuntimeError, NotImplementedError):  # backends without the fused form
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat, group=group)
        out = torch.cat(parts)
    if out.dtype != dtype:
        out = out.view(dtype)
    return out.reshape((world,) + shape)


def all_gather_hits(hits, group=None):
    """All-gather the per-rank hit lists: [nq, L, 32] uint8 -> [world, nq, L, 32] (same device)."""
    return all_gather_stack(hits, group)


def exchange_packed(packed, cnt, group=None):
    """The packed exchange: every rank contributes its valid hits only.  packed [>= total, 4] int64 (cis_hit records of
    this rank in query order, first `total` rows valid), cnt [nq] int32.  Returns (parts [world, stride, 4],
    off [world, nq] int64, cnt_all [world, nq] int32) with stride = the largest per-rank total (one host read).
    Works on CPU tensors with the gloo backend as well."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    nq = int(cnt.shape[0])
    cnt_all = all_gather_stack(cnt, group)
    csum = torch.cumsum(cnt_all, dim=1, dtype=torch.int64)
    stride = max(int(csum[:, -1].max().item()), 1) if nq else 1
    mine = packed[:stride]
    if mine.shape[0] < stride:  # a buffer sized for this rank's own total only
        mine = torch.cat([mine, torch.zeros((stride - mine.shape[0], 4), dtype=packed.dtype, device=packed.device)])
    parts = all_gather_stack(mine.contiguous(), group)
    return parts, (csum - cnt_all).contiguous(), cnt_all


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

    def search_batch_dev(self, q, quota=10, limit=None, packed=True):
        """packed=True (default): only valid hits travel.  A rank owns 1/world of the cells, so its [nq, L] partial
        list is mostly empty slots; the dense all-gather moves world * nq * L * 32 B to every rank (26 MB per rank at
        8192 x 100), the packed one about nq * L * 32 B in total (+ 4 B per query and rank of counts)."""
        import torch
        import torch.distributed as dist
        from .lopq.search import merge_hits_dev, merge_packed_dev
        L = self.local._dev_args(q, quota, limit)[0]
        if not packed or L > 512 or L == 0:
            hits, visited = self.local.search_partial_dev(q, quota=quota, limit=limit)
            out = merge_hits_dev(all_gather_hits(hits, self.group))
            out["visited"] = visited
            return out
        nq = int(q.shape[0])
        p = self.local.search_partial_packed_dev(q, quota=quota, limit=limit)
        parts, off, cnt_all = exchange_packed(p["packed"], p["cnt"], self.group)
        out = merge_packed_dev(parts, off, cnt_all, nq, L)
        out["visited"] = p["visited"]
        return out
