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


def merge_packed_sorted(parts, off, cnt, nq, L):
    """Merge packed per-shard hit lists for ANY limit: parts [world, stride, 4] int64 (cis_hit records: dist bits,
    visit_rank | pos << 32, id, cell), off [world, nq] int64, cnt [world, nq] int32 -> dict like merge_packed_dev.
    The ranking key (dist, visit_rank, pos) of lopq/lopq/search.py:128-133,:210 is applied as three stable device
    sorts (least significant key first), then the first L records of every query are scattered out.  Used above the
    512 records per query that cis_merge_packed_dev ranks in one wave."""
    import torch
    world, stride = int(parts.shape[0]), int(parts.shape[1])
    dev = parts.device
    out = {"ids": torch.full((nq, L), -1, dtype=torch.int64, device=dev),
           "dists": torch.full((nq, L), float("nan"), dtype=torch.float64, device=dev),
           "n_found": torch.zeros(nq, dtype=torch.int32, device=dev)}
    if nq == 0 or L == 0:
        return out
    cnt64 = cnt.to(torch.int64)
    # valid records of every (shard, query): positions off .. off + cnt in that shard's packed array
    j = torch.arange(stride, device=dev, dtype=torch.int64)
    end = (off + cnt64)                                         # [world, nq]
    last = end[:, -1:]                                          # records of a shard that are valid at all
    valid = j[None, :] < last                                   # [world, stride]
    # query of a record: number of queries whose range ends at or before it
    qid = torch.searchsorted(end.contiguous(), j[None, :].expand(world, stride).contiguous(), right=True)
    rec = parts[valid]                                          # [T, 4]
    q = qid[valid]
    dist_bits = rec[:, 0]                                       # positive finite doubles order like their bit patterns
    w1 = rec[:, 1]
    key_rp = ((w1 & 0xFFFFFFFF) << 32) | ((w1 >> 32) & 0xFFFFFFFF)   # visit_rank (low word) major, pos (high word) minor
    order = torch.sort(key_rp, stable=True).indices
    order = order[torch.sort(dist_bits[order], stable=True).indices]
    order = order[torch.sort(q[order], stable=True).indices]
    qs = q[order]
    tot = torch.bincount(qs, minlength=nq)                      # merged candidates per query
    start = torch.cumsum(tot, 0) - tot
    k = torch.arange(qs.shape[0], device=dev, dtype=torch.int64) - start[qs]
    keep = k < L
    sel, qk, kk = order[keep], qs[keep], k[keep]
    out["ids"][qk, kk] = rec[sel, 2]
    out["dists"][qk, kk] = dist_bits[sel].view(torch.float64)
    out["n_found"] = torch.clamp(tot, max=L).to(torch.int32)
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

    def search_batch_dev(self, q, quota=10, limit=None, packed=True):
        """packed=True (default): only valid hits travel.  A rank owns 1/world of the cells, so its [nq, L] partial
        list is mostly empty slots; the dense all-gather moves world * nq * L * 32 B to every rank (26 MB per rank at
        8192 x 100), the packed one about nq * L * 32 B in total (+ 4 B per query and rank of counts)."""
        import torch
        import torch.distributed as dist
        from .lopq.search import merge_hits_dev, merge_packed_dev
        L = self.local._dev_args(q, quota, limit)[0]
        if (not packed and L <= 3072) or L == 0:
            hits, visited = self.local.search_partial_dev(q, quota=quota, limit=limit)
            out = merge_hits_dev(all_gather_hits(hits, self.group))
            out["visited"] = visited
            return out
        nq = int(q.shape[0])
        p = self.local.search_partial_packed_dev(q, quota=quota, limit=limit)
        parts, off, cnt_all = exchange_packed(p["packed"], p["cnt"], self.group)
        if L <= 512:
            out = merge_packed_dev(parts, off, cnt_all, nq, L)      # one wave per query
        else:
            out = merge_packed_sorted(parts, off, cnt_all, nq, L)   # any limit: stable device sorts
        out["visited"] = p["visited"]
        return out
