"""Test-side restatement of the packed any-limit merge with torch sorts (it used to be the product path of limit > 3072;
the product merges with k_merge_packed_ranked now).  Used by the gloo CPU test and as the checker of the GPU test."""


def merge_packed_sorted(parts, off, cnt, nq, L):
    """Merge packed per-shard hit lists for ANY limit: parts [world, stride, 4] int64 (cis_hit records: dist bits,
    visit_rank | pos << 32, id, cell), off [world, nq] int64, cnt [world, nq] int32 -> dict like merge_packed_dev.
    The ranking key (dist, visit_rank, pos) of lopq/lopq/search.py:128-133,:210 is applied as three stable device
    sorts (least significant key first), then the first L records of every query are scattered out.  Used above the
    3072 records per query that cis_merge_packed_dev ranks in one wave."""
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


def pack_hits_dev(hits):
    """Valid hits of a partial result [nq, L, 32] (uint8) packed in query order: (packed [total, 4] int64, cnt [nq] int32).
    Valid hits are a prefix of every row (id >= 0).  The boolean indexing synchronises with the device."""
    import torch
    nq, L = int(hits.shape[0]), int(hits.shape[1])
    hv = hits.view(torch.int64).view(nq, L, 4)
    valid = hv[:, :, 2] >= 0
    return hv[valid], valid.sum(dim=1, dtype=torch.int32)
