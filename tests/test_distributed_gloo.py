"""N > 1 path on CPU: two gloo processes run the cell-sharding protocol of
columbiaimagesearch_amd/distributed.py with the oracle standing in for the GPU scan.

Checks (a) the product's all_gather_hits plumbing on CPU tensors, (b) that shards built from the
replicated cell-size table agree on `visited`, and (c) that merging the per-shard lists by
(dist, visit_rank, pos) reproduces the single-index ranking exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, quota, limit, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from columbiaimagesearch_amd.distributed import all_gather_hits, greedy_cell_owner
        from oracle import lopq_oracle as O
        z, X, Q = load_golden(name)
        m = O.OracleModel.from_npz(z)
        ix = O.OracleCSRIndex(m, z["coarse"], z["fine"])
        V = m.V
        counts = np.diff(ix.offsets)
        owner = greedy_cell_owner(counts, world)
        nq = 12
        hits = np.zeros((nq, limit), dtype=O.HIT_DTYPE)
        vis = np.zeros(nq, dtype=np.int64)
        for qi in range(nq):
            hits[qi], vis[qi] = O.search_partial(ix, Q[qi], quota, limit, owner, rank)
        t = torch.from_numpy(hits.view(np.uint8).reshape(nq, limit, 32).copy())
        parts = all_gather_hits(t).numpy().reshape(world, nq, limit * 32).view(O.HIT_DTYPE).reshape(world, nq, limit)
        ok = True
        for qi in range(nq):
            merged = O.merge_partials([parts[w, qi] for w in range(world)], limit)
            ids, dists, visited = ix.search(Q[qi], quota=quota, limit=limit)
            ok = ok and visited == vis[qi] and np.array_equal(merged["id"], ids) and np.array_equal(merged["dist"], dists)
        # the packed exchange (valid hits only) carries the same lists
        from columbiaimagesearch_amd.distributed import exchange_packed
        hv = torch.from_numpy(hits.view(np.int64).reshape(nq, limit, 4).copy())
        valid = hv[:, :, 2] >= 0
        pparts, off, cnt_all = exchange_packed(hv[valid], valid.sum(dim=1, dtype=torch.int32))
        for w in range(world):
            for qi in range(nq):
                c, o = int(cnt_all[w, qi]), int(off[w, qi])
                lst = pparts[w, o:o + c].numpy().view(O.HIT_DTYPE).reshape(-1)
                ref_lst = parts[w, qi][parts[w, qi]["id"] >= 0]
                ok = ok and np.array_equal(lst, ref_lst)
        # the any-limit merge of the packed lists (stable sorts by pos/visit_rank, dist, query) reproduces the single index
        from ref_merge import merge_packed_sorted
        mo = merge_packed_sorted(pparts, off, cnt_all, nq, limit)
        for qi in range(nq):
            ids, dists, _ = ix.search(Q[qi], quota=quota, limit=limit)
            k = int(mo["n_found"][qi])
            ok = ok and k == len(ids) and np.array_equal(mo["ids"][qi, :k].numpy(), ids)
            ok = ok and np.array_equal(mo["dists"][qi, :k].numpy(), dists) and bool((mo["ids"][qi, k:] == -1).all())
        # 16-bit coarse codes travel as bytes (neither RCCL nor gloo has an int16 collective)
        from columbiaimagesearch_amd.distributed import all_gather_stack
        c16 = (torch.arange(6, dtype=torch.int16).reshape(3, 2) + 1000 * rank)
        g16 = all_gather_stack(c16)
        ok = ok and g16.dtype == torch.int16 and tuple(g16.shape) == (world, 3, 2)
        ok = ok and all(torch.equal(g16[w], torch.arange(6, dtype=torch.int16).reshape(3, 2) + 1000 * w) for w in range(world))
        allv = [torch.zeros(nq, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allv, torch.from_numpy(vis))
        ok = ok and all(torch.equal(allv[0], v) for v in allv)
        owned = int((owner == rank).sum())
        ret[rank] = (ok, owned)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("quota,limit", [(50, 20), (3000, 100), (3000, 700)])
def test_cell_sharded_search_two_gloo_ranks(quota, limit):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "c2", quota, limit, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret[r][0] for r in range(world))
    assert ret[0][1] + ret[1][1] == 256 and min(ret[0][1], ret[1][1]) > 0


def test_greedy_owner_is_balanced_and_deterministic():
    from columbiaimagesearch_amd.distributed import greedy_cell_owner
    rs = np.random.RandomState(0)
    counts = rs.randint(0, 100000, size=256)
    for world in (2, 4, 8):
        o1, o2 = greedy_cell_owner(counts, world), greedy_cell_owner(counts.copy(), world)
        assert np.array_equal(o1, o2)
        load = np.bincount(o1, weights=counts, minlength=world)
        assert load.max() <= 1.05 * load.mean()


def _route_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from columbiaimagesearch_amd.distributed import greedy_cell_owner, route_codes
        z, X, Q = load_golden("c2")
        coarse, fine = z["coarse"][:9000], z["fine"][:9000]
        n, V = coarse.shape[0], 16
        ids = np.arange(n, dtype=np.int64) * 3 + 1
        cell = coarse[:, 0].astype(np.int64) * V + coarse[:, 1]
        owner = greedy_cell_owner(np.bincount(cell, minlength=V * V), world)
        a, b = rank * n // world, (rank + 1) * n // world
        c, f, i = route_codes(coarse[a:b], fine[a:b], ids[a:b], owner, V)
        mine = owner[cell] == rank  # what a single pass over the concatenated batch hands this rank, in batch order
        ok = np.array_equal(c, coarse[mine]) and np.array_equal(f, fine[mine]) and np.array_equal(i, ids[mine])
        e = route_codes(coarse[:0], fine[:0], ids[:0], owner, V)  # an empty slice on every rank
        ok = ok and e[0].shape == (0, 2) and e[1].shape[0] == 0 and e[2].shape == (0,)
        # ONE rank with nothing to bring, handed as a 1-D empty array: the record width comes from the model's M, not from
        # the local array (a rank that derived M = 0 would announce 12-byte records to peers sending 12 + M)
        M = fine.shape[1]
        if rank == 1:
            c2, f2, i2 = route_codes(np.zeros((0, 2), np.uint16), np.zeros(0, np.uint8), np.zeros(0, np.int64), owner, V, M=M)
        else:
            c2, f2, i2 = route_codes(coarse[a:b], fine[a:b], ids[a:b], owner, V, M=M)
        keep = mine & ~((np.arange(n) >= n // world) & (np.arange(n) < 2 * n // world))  # rank 1's slice is missing
        ok = ok and np.array_equal(c2, coarse[keep]) and np.array_equal(f2, fine[keep]) and np.array_equal(i2, ids[keep])
        ret[rank] = (bool(ok), int(mine.sum()))
    finally:
        dist.destroy_process_group()


def test_routed_insert_all_to_all_three_gloo_ranks():
    """every code travels once, to the owner of its cell, and arrives in batch order."""
    world = 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_route_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret[r][0] for r in range(world))
    assert sum(ret[r][1] for r in range(world)) == 9000


def _grid_worker(rank, world, port, S, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from columbiaimagesearch_amd.distributed import all_gather_rows, greedy_cell_owner, grid_groups, route_codes
        z, X, Q = load_golden("c2")
        coarse, fine = z["coarse"][:9000], z["fine"][:9000]
        n, V = coarse.shape[0], 16
        ids = np.arange(n, dtype=np.int64) * 3 + 1
        cell = coarse[:, 0].astype(np.int64) * V + coarse[:, 1]
        g, s, R, S_, row, col = grid_groups(S)
        ok = (g, s, R, S_) == (rank // S, rank % S, world // S, S)
        owner = greedy_cell_owner(np.bincount(cell, minlength=V * V), S)
        k = s * R + g  # GridSearcher.slice_number
        cuts = [0] + [(j + 1) * n // world + 13 * (j + 1) - 40 for j in range(world - 1)] + [n]  # ragged slices
        a, b = cuts[k], cuts[k + 1]
        # column: the ranks holding the same cells in the other copies (the whole world when S == 1)
        cg = col if S > 1 else None
        if R > 1:
            c = all_gather_rows(torch.from_numpy(coarse[a:b].view(np.int16).copy()), cg).numpy().view(np.uint16)
            f = all_gather_rows(torch.from_numpy(fine[a:b].copy()), cg).numpy()
            i = all_gather_rows(torch.from_numpy(ids[a:b].copy()), cg).numpy()
        else:
            c, f, i = coarse[a:b], fine[a:b], ids[a:b]
        # after the column gather rank (g, s) holds slices s*R .. s*R+R-1: a contiguous range of the data
        ok = ok and np.array_equal(i, ids[cuts[s * R]:cuts[s * R + R]]) and np.array_equal(c, coarse[cuts[s * R]:cuts[s * R + R]])
        if S > 1:  # row: the S ranks of one copy
            c, f, i = route_codes(c, f, i, owner, V, group=row if R > 1 else None, M=fine.shape[1])
        mine = owner[cell] == s
        ok = ok and np.array_equal(c, coarse[mine]) and np.array_equal(f, fine[mine]) and np.array_equal(i, ids[mine])
        ret[rank] = (bool(ok), int(mine.sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S", [1, 2, 4])
def test_grid_build_four_gloo_ranks(S):
    """R query groups x S cell shards at world 4: after the ragged column all-gather and the routed insert inside the row,
    every copy's shard s holds exactly the items of its cells in the order of the whole data (GridSearcher's build)."""
    world = 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_grid_worker, args=(r, world, port, S, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret[r][0] for r in range(world))
    assert sum(ret[r][1] for r in range(world)) == 9000 * (world // S)


def _routed_worker(rank, world, port, quota, limit, cap, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from columbiaimagesearch_amd.distributed import (greedy_cell_owner, home_slice, route_slots_torch, route_rows_torch,
                                                          routed_send_queries, routed_return_hits, routed_merge_tables)
        from oracle import lopq_oracle as O
        z, X, Q = load_golden("c2")
        m = O.OracleModel.from_npz(z)
        ix = O.OracleCSRIndex(m, z["coarse"], z["fine"])
        owner = greedy_cell_owner(np.diff(ix.offsets), world)
        nq = 14
        lo, hi = home_slice(nq, rank, world)
        qh = np.ascontiguousarray(Q[lo:hi], dtype=np.float32)
        # home: owners of the visited cells
        mv = [O.query_owners(ix, q, quota, owner) for q in qh]
        mask = torch.tensor([a for a, _ in mv], dtype=torch.int64)
        slot, cnt, ov = route_slots_torch(mask, world, cap)
        send = route_rows_torch(torch.from_numpy(qh), slot, cap)
        # out: queries to the owners
        recv_q, recv_cnt, ov = routed_send_queries(torch.from_numpy(qh), slot, send, cnt, ov)
        if int(ov.item()):  # a block overflowed on SOME rank: every rank sees the flag
            ret[rank] = ("overflow", int(cnt.max()))
            return
        n_sent, n_recv = cnt.tolist(), recv_cnt.tolist()
        rows = torch.cat([recv_q[s, :n_recv[s]] for s in range(world)]).numpy()
        # scan: the oracle answers for this rank's cells
        hits = np.zeros((rows.shape[0], limit), dtype=O.HIT_DTYPE)
        for i in range(rows.shape[0]):
            hits[i], _ = O.search_partial(ix, rows[i], quota, limit, owner, rank)
        ht = torch.from_numpy(hits.view(np.uint8).reshape(rows.shape[0], limit, 32).copy())
        # back + merge tables
        back = routed_return_hits(ht, n_recv, n_sent)
        rec = back.reshape(-1).view(torch.int64).reshape(-1, 4)
        valid = (rec[:, 2].reshape(-1, limit) >= 0).sum(dim=1, dtype=torch.int32)
        off, c2 = routed_merge_tables(slot, n_sent, valid, limit)
        flat = back.numpy().reshape(-1).view(O.HIT_DTYPE)
        ok = True
        asked = 0
        for i in range(hi - lo):
            lists = [flat[int(off[d, i]):int(off[d, i]) + int(c2[d, i])] for d in range(world)]
            asked += sum(1 for d in range(world) if int(slot[d, i]) >= 0)
            merged = O.merge_partials(lists, limit) if sum(len(l) for l in lists) else np.zeros(0, dtype=O.HIT_DTYPE)
            ids, dists, visited = ix.search(qh[i], quota=quota, limit=limit)
            ok = ok and visited == mv[i][1] and np.array_equal(merged["id"], ids) and np.array_equal(merged["dist"], dists)
        ret[rank] = (ok, asked, hi - lo, rows.shape[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("quota,limit", [(50, 20), (3000, 100)])
def test_routed_search_three_gloo_ranks(quota, limit):
    """The routed protocol of columbiaimagesearch_amd/distributed.py (home -> owners -> home) on CPU tensors, the oracle standing in
    for the GPU kernels: every home slice equals the single index, and a query travels to the owners of its cells only."""
    world = 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_routed_worker, args=(r, world, port, quota, limit, 8, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret[r][0] is True for r in range(world)), dict(ret)
    asked, homes, received = sum(ret[r][1] for r in range(world)), sum(ret[r][2] for r in range(world)), sum(ret[r][3] for r in range(world))
    assert homes == 14 and asked == received and homes <= asked < homes * world  # owners only, not everybody


def test_routed_search_overflow_flag_is_seen_by_every_rank():
    world = 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_routed_worker, args=(r, world, port, 3000, 10, 1, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret[r][0] == "overflow" for r in range(world)), dict(ret)


def test_routing_tables_against_a_brute_force():
    """route_slots_torch / route_rows_torch / routed_merge_tables (the restatements the HIP kernels are checked against on the GPU)
    against plain loops: stable compaction per destination, capacity cut + overflow flag, offsets and counts of the returned lists."""
    from columbiaimagesearch_amd.distributed import home_slice, route_capacity, route_rows_torch, route_slots_torch, routed_merge_tables
    rs = np.random.RandomState(4)
    for world, nq, cap in [(1, 5, 5), (3, 17, 17), (8, 64, 9), (5, 0, 4), (64, 33, 33)]:
        mask = rs.randint(0, 1 << min(world, 62), size=nq, dtype=np.int64) if nq else np.zeros(0, dtype=np.int64)
        if world == 64 and nq:
            mask[0] = -1  # all 64 bits
        slot, cnt, ov = route_slots_torch(torch.from_numpy(mask), world, cap)
        q = torch.arange(nq * 3, dtype=torch.float32).reshape(nq, 3)
        rows = route_rows_torch(q, slot, cap)
        over = False
        for d in range(world):
            want = [i for i in range(nq) if (int(mask[i]) >> d) & 1]
            over = over or len(want) > cap
            assert int(cnt[d]) == min(len(want), cap)
            for pos, i in enumerate(want):
                assert int(slot[d, i]) == (pos if pos < cap else -1)
                if pos < cap:
                    assert torch.equal(rows[d, pos], q[i])
            assert all(int(slot[d, i]) == -1 for i in range(nq) if i not in want)
        assert bool(ov.item()) == over
        # the return trip: rank d answers cnt[d] rows; row r of its group holds valid[...] hits
        n_sent = cnt.tolist()
        L = 7
        valid = torch.from_numpy(rs.randint(0, L + 1, size=int(sum(n_sent))).astype(np.int32))
        off, c = routed_merge_tables(slot, n_sent, valid, L)
        base = np.concatenate([[0], np.cumsum(n_sent)[:-1]]) if world else []
        for d in range(world):
            for i in range(nq):
                sl = int(slot[d, i])
                if sl < 0:
                    assert int(c[d, i]) == 0
                else:
                    assert int(off[d, i]) == (int(base[d]) + sl) * L and int(c[d, i]) == int(valid[int(base[d]) + sl])
    assert [home_slice(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    assert route_capacity(100, 32, 8) == 100 and route_capacity(100, 1024, 8) < 100 and route_capacity(0, 32, 8) == 1
