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


def exchange_stride(nq, L, world, slack=1.5):
    """Fixed per-rank size (records) of the payload all-gather: a rank owns 1 / world of the cells, so it holds about nq * L / world
    of the batch's hits; `slack` times that (never more than nq * L, what a rank can hold at all).  A rank above it raises the
    overflow flag and the exchange is repeated with the exact size."""
    full = int(nq) * int(L)
    if world <= 1:
        return max(full, 1)
    return max(1, min(full, int(full / float(world) * slack) + 1024))


def exchange_packed_fixed(packed, cnt, stride, group=None):
    """The packed exchange WITHOUT a host read (round 4): the payload all-gather moves a fixed `stride` records per rank
    (exchange_stride), the offsets come from a kernel (cis_exchange_offsets_dev) and a device-side flag says whether some rank
    held more than `stride` records.  Device tensors only.  Returns (parts [world, stride, 4], off [world, nq] int64,
    cnt_all [world, nq] int32, overflow int32 [1])."""
    import torch
    import torch.distributed as dist
    from . import _lib
    world = dist.get_world_size(group)
    nq = int(cnt.shape[0])
    dev = cnt.device
    cnt_all = all_gather_stack(cnt, group)
    off = torch.empty((world, nq), dtype=torch.int64, device=dev)
    totals = torch.empty(world, dtype=torch.int64, device=dev)
    overflow = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().cis_exchange_offsets_dev(cnt_all.data_ptr(), world, nq, int(stride), off.data_ptr(), totals.data_ptr(),
                                                   overflow.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    if packed.shape[0] < stride:
        raise ValueError("the packed buffer holds %d records, the exchange stride is %d" % (packed.shape[0], stride))
    parts = all_gather_stack(packed[:stride].contiguous(), group)
    return parts, off, cnt_all, overflow


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
    if cnt_all.is_cuda and nq:
        # offsets and per-rank totals by the kernel of the fixed-size exchange (round 5: this path used torch.cumsum); the ONE host read
        # of this exact-size form is the largest total -- it is only taken after a fixed-size exchange overflowed
        from . import _lib
        off = torch.empty((world, nq), dtype=torch.int64, device=cnt.device)
        totals = torch.empty(world, dtype=torch.int64, device=cnt.device)
        flag = torch.empty(1, dtype=torch.int32, device=cnt.device)
        _lib.check(_lib.lib().cis_exchange_offsets_dev(cnt_all.data_ptr(), world, nq, (1 << 62), off.data_ptr(), totals.data_ptr(),
                                                       flag.data_ptr(), torch.cuda.current_stream(cnt.device).cuda_stream))
        stride = max(int(totals.max().item()), 1)
    else:
        csum = torch.cumsum(cnt_all, dim=1, dtype=torch.int64)
        stride = max(int(csum[:, -1].max().item()), 1) if nq else 1
        off = (csum - cnt_all).contiguous()
    if packed.shape[0] >= stride:
        mine = packed[:stride]
    else:  # a buffer sized for this rank's own total only
        mine = torch.zeros((stride, 4), dtype=packed.dtype, device=packed.device)
        mine[:packed.shape[0]] = packed
    parts = all_gather_stack(mine.contiguous(), group)
    return parts, off, cnt_all


def route_codes(coarse, fine, ids, owner, V, group=None, M=None):
    """All-to-all routing of freshly encoded codes to the ranks that own their cells.
    coarse [n,2] uint16, fine [n,M] uint8, ids [n] int64: THIS rank's slice of the batch (host arrays).  Returns the
    (coarse, fine, ids) this rank owns, ordered by source rank and, inside a source, in the source's order -- i.e. in
    the order of the concatenated batch, so that the per-cell insertion order (= the reference's list order,
    lopq/lopq/search.py:359) is the one a single index would have.  One record is 8 + 4 + M bytes; one collective for
    the split sizes, one for the records (RCCL send/recv groups on GPUs; the same calls run on gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    coarse = np.ascontiguousarray(np.asarray(coarse).reshape(-1, 2), dtype=np.uint16)
    n = coarse.shape[0]
    fine = np.asarray(fine)
    if M is None:  # the record width must be the same on every rank: callers with possibly empty slices pass the model's M
        M = int(fine.shape[-1]) if fine.ndim == 2 else (fine.size // n if n else 0)
    if fine.size != n * M:
        raise ValueError("fine must hold %d x %d codes" % (n, M))
    fine = np.ascontiguousarray(fine.reshape(n, M), dtype=np.uint8)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    rec = np.zeros((n, 12 + M), dtype=np.uint8)
    rec[:, :8] = ids.view(np.uint8).reshape(n, 8)
    rec[:, 8:12] = coarse.view(np.uint8).reshape(n, 4)
    rec[:, 12:] = fine
    dst = np.asarray(owner)[coarse[:, 0].astype(np.int64) * V + coarse[:, 1]] if n else np.zeros(0, dtype=np.int64)
    order = np.argsort(dst, kind="stable")  # records grouped by destination, original order inside a group
    send_counts = np.bincount(dst, minlength=world).astype(np.int64)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    sc = torch.from_numpy(send_counts).to(dev)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = rc.cpu().numpy()
    send = torch.from_numpy(rec[order].reshape(-1)).to(dev)
    recv = torch.empty(int(recv_counts.sum()) * (12 + M), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=[int(c) * (12 + M) for c in recv_counts],
                           input_split_sizes=[int(c) * (12 + M) for c in send_counts], group=group)
    got = recv.cpu().numpy().reshape(-1, 12 + M)
    return (np.ascontiguousarray(got[:, 8:12]).view(np.uint16).reshape(-1, 2), np.ascontiguousarray(got[:, 12:]),
            np.ascontiguousarray(got[:, :8]).view(np.int64).reshape(-1))


class ShardedSearcher(object):
    """LOPQSearcherHIP sharded by coarse cell over the ranks of a torch.distributed group."""

    def __init__(self, model, owner=None, group=None):
        import torch.distributed as dist
        from .lopq.search import LOPQSearcherHIP
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.local = LOPQSearcherHIP(model, shard=(self.rank, self.world, owner))
        self._owner = None if owner is None else np.ascontiguousarray(owner, dtype=np.int32)
        self._side = None

    # -- inserts against searches in flight -----------------------------------------------------------------------------------
    # search_begin runs partial searches on this object's own lane streams over VIEWS of the local index (lanes()); an insert
    # rewrites the arrays those kernels read (in place: lend / gcount / ids / codes; a rebuild swaps the generations).  The caller
    # cannot see the lane streams, so every insert entry point orders itself against them: the insert starts after the searches
    # that were begun before it, and searches begun after it start after the insert.
    def _lane_streams(self):
        return [s for _, s in (getattr(self, "_lanes", None) or []) if s is not None]

    def _insert_fence(self, host=False):
        """Before an insert.  host=True: the host-pointer entry points run on the library's own stream -- wait on the host."""
        import torch
        streams = self._lane_streams()
        if not streams:
            return
        if host:
            torch.cuda.current_stream().synchronize()
            for s in streams:
                s.synchronize()
        else:
            cur = torch.cuda.current_stream()
            for s in streams:
                cur.wait_stream(s)

    def _insert_release(self):
        """After an insert: later searches on the lane streams wait for it."""
        import torch
        cur = torch.cuda.current_stream()
        for s in self._lane_streams():
            s.wait_stream(cur)

    def add_codes_array(self, coarse, fine, ids=None, dedup=True):
        """Every rank is given ALL codes (it keeps its own cells and counts the rest)."""
        self._insert_fence(host=True)
        return self.local.add_codes_array(coarse, fine, ids, dedup)

    def add_codes_dev(self, coarse, fine, ids, dedup=True):
        """add_codes_array on tensors in HBM (every rank is given ALL codes), ordered against the searches in flight."""
        self._insert_fence()
        try:
            return self.local.add_codes_dev(coarse, fine, ids, dedup=dedup)
        finally:
            self._insert_release()

    def close(self):
        """The views of the lanes first (they share the local index's storage), then the local index."""
        import torch
        for sv, s in (getattr(self, "_lanes", None) or [])[1:]:
            if s is not None:
                s.synchronize()
            sv.close()
        self._lanes = None
        if self.local is not None:
            self.local.close()

    def add_codes_routed(self, coarse, fine, ids, dedup=True):
        """Every rank brings ITS slice of a batch (e.g. what it encoded itself); the codes travel once, to the owner of
        their cell (all-to-all), the owners insert them, and the accepted per-cell counts are summed over the ranks so
        that every rank ends with the whole cell-size table.  Equivalent to add_codes_array of the concatenated batch
        (rank 0's slice first) on every rank, at 1/world of the traffic and host work."""
        import torch
        import torch.distributed as dist
        from . import _lib
        L = _lib.lib()
        V = self.local.model.V
        owner = self._owner if self._owner is not None else np.arange(V * V) % self.world
        self._insert_fence(host=True)
        c, f, i = route_codes(coarse, fine, ids, owner, V, self.group, M=self.local._M)
        before = np.zeros(V * V, dtype=np.int64)
        _lib.check(L.cis_index_cell_counts(self.local._ix, _lib.ptr(before)))
        added = self.local.add_codes_array(c, f, i, dedup) if c.shape[0] else 0
        after = np.zeros(V * V, dtype=np.int64)
        _lib.check(L.cis_index_cell_counts(self.local._ix, _lib.ptr(after)))
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        delta = torch.from_numpy(after - before).to(dev)
        dist.all_reduce(delta, group=self.group)
        delta = np.ascontiguousarray(delta.cpu().numpy())
        _lib.check(L.cis_index_add_remote_counts(self.local._ix, _lib.ptr(delta)))
        self.local.nb_indexed = int(L.cis_index_size(self.local._ix))
        return added

    def add_codes_routed_dev(self, coarse, fine, ids, dedup=True):
        """add_codes_routed on tensors in HBM (this rank's freshly encoded slice: coarse [n,2] 16-bit, fine [n,M] uint8,
        ids [n] int64).  The records are grouped by owner on the device (cis_index_route_pack_dev), travel once in ONE
        all-to-all of device buffers (RCCL over xGMI; only the world split sizes are read by the host, which the collective's
        interface needs), are merged into the owner's HBM index by kernels (cis_index_add_records_dev), and the per-cell
        accepted counts are all-reduced on the device.  Returns the number of items this rank accepted."""
        import torch
        import torch.distributed as dist
        from . import _lib
        L = _lib.lib()
        ix = self.local._ix
        M, V = self.local._M, self.local.model.V
        n = int(coarse.shape[0])
        if tuple(fine.shape) != (n, M) or tuple(ids.shape) != (n,) or tuple(coarse.shape) != (n, 2):
            raise ValueError("coarse [n,2], fine [n,%d] and ids [n] expected" % M)
        dev = coarse.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        self._insert_fence()  # the partial searches begun before this call read the arrays the insert rewrites
        try:   # (an exception from _lib.check or a collective must not leave later lane-stream searches unordered against a partly enqueued insert)
            rec_b = 12 + M
            send = torch.empty(max(n, 1) * rec_b, dtype=torch.uint8, device=dev)
            sc = torch.empty(self.world, dtype=torch.int64, device=dev)
            _lib.check(L.cis_index_route_pack_dev(ix, ids.data_ptr(), coarse.data_ptr(), fine.data_ptr(), n, send.data_ptr(),
                                                  sc.data_ptr(), stream))
            staged = dist.get_backend(self.group) != "nccl"  # gloo has no all-to-all on device buffers: stage the collectives
            rc = torch.empty_like(sc)
            if staged:
                sc_h = sc.cpu()
                rc_h = torch.empty_like(sc_h)
                dist.all_to_all_single(rc_h, sc_h, group=self.group)
            else:
                dist.all_to_all_single(rc, sc, group=self.group)
                sc_h, rc_h = sc.cpu(), rc.cpu()
            in_split = [int(c) * rec_b for c in sc_h.tolist()]
            out_split = [int(c) * rec_b for c in rc_h.tolist()]
            n_recv = sum(out_split) // rec_b
            recv = torch.empty(max(n_recv, 1) * rec_b, dtype=torch.uint8, device=dev)
            if staged:
                recv_h = torch.empty(n_recv * rec_b, dtype=torch.uint8)
                dist.all_to_all_single(recv_h, send[:n * rec_b].cpu(), output_split_sizes=out_split, input_split_sizes=in_split,
                                       group=self.group)
                recv[:n_recv * rec_b].copy_(recv_h)
            else:
                dist.all_to_all_single(recv[:n_recv * rec_b], send[:n * rec_b], output_split_sizes=out_split,
                                       input_split_sizes=in_split, group=self.group)
            delta = torch.empty(V * V, dtype=torch.int64, device=dev)
            added, bad = _lib.c_int64(0), _lib.c_int64(0)
            _lib.check(L.cis_index_add_records_dev(ix, recv.data_ptr(), n_recv, 1 if dedup else 0, _lib.ctypes.byref(added),
                                                   _lib.ctypes.byref(bad), delta.data_ptr(), stream))
            if staged:
                d_h = delta.cpu()
                dist.all_reduce(d_h, group=self.group)
                delta.copy_(d_h)
            else:
                dist.all_reduce(delta, group=self.group)
            _lib.check(L.cis_index_add_remote_counts_dev(ix, delta.data_ptr(), stream))
        finally:
            self._insert_release()
        self.local.nb_indexed = int(L.cis_index_size(ix))
        if bad.value:
            print("Could not push {} codes (out of range for this model, or negative ids).".format(bad.value))
        return int(added.value)

    def get_nb_indexed(self):
        return self.local.get_nb_indexed()

    # -- pipelined form: the exchange + merge of batch b run on a side stream while the caller launches batch b+1 --------
    pipeline_depth = 3   # partial searches in flight: consecutive search_begin calls rotate over this many views of the local index

    def lanes(self):
        """[(searcher, stream or None)]: the local index and its views (cis_index_create_view: shared storage, own workspaces), each
        on its own stream, so that the small kernels of one batch's front end overlap another batch's scan."""
        import torch
        if getattr(self, "_lanes", None) is None:
            from .streams import lane_streams   # (streams on different hardware pipes, made once per process: streams.py)
            depth = max(1, int(self.pipeline_depth))
            self._lanes = [(self.local, None)]
            for ls in lane_streams(depth - 1):
                self._lanes.append((self.local.view(), ls))
            self._turn = 0
        return self._lanes

    def _lane(self):
        """(searcher, stream) of the next search_begin."""
        lanes = self.lanes()
        lane = lanes[self._turn % len(lanes)]
        self._turn += 1
        return lane

    def search_begin(self, q, quota=10, limit=None):
        """This rank's partial search (asynchronous).  Returns a handle for search_end.  Consecutive calls run on different lanes
        (see _lane): the caller's current stream for the first, side streams that wait for the caller's stream for the others."""
        import torch
        L = self.local._dev_args(q, quota, limit)[0]
        sv, stream = self._lane()
        if stream is None:
            p = sv.search_partial_packed_dev(q, quota=quota, limit=limit)
            ev = torch.cuda.Event()
            ev.record()
        else:
            stream.wait_stream(torch.cuda.current_stream())  # the queries were produced on the caller's stream
            q.record_stream(stream)
            with torch.cuda.stream(stream):
                p = sv.search_partial_packed_dev(q, quota=quota, limit=limit)
                ev = torch.cuda.Event()
                ev.record(stream)
        return {"p": p, "ev": ev, "nq": int(q.shape[0]), "L": L, "searcher": sv}

    def search_end(self, h, check=True):
        """Exchange + merge of a search_begin handle on the side stream; the result tensors are safe to use on the
        current stream when this returns (it waits for the side stream's event, not for the device).  check=False: no host
        read at all -- the caller verifies `overflowed(outs)` once after its loop."""
        import torch
        if self._side is None:
            from .streams import lane_streams   # the lane after the search lanes': another pipe than theirs while there are four or fewer
            self._side = lane_streams(max(1, int(self.pipeline_depth)))[-1]
        cur = torch.cuda.current_stream()
        p = h["p"]
        with torch.cuda.stream(self._side):
            self._side.wait_event(h["ev"])
            for t in (p["packed"], p["cnt"], p["visited"]):
                t.record_stream(self._side)
            out = self._exchange_and_merge(p, h["nq"], h["L"], check=check)
            done = torch.cuda.Event()
            done.record(self._side)
        for t in out.values():
            if hasattr(t, "record_stream"):
                t.record_stream(cur)
        cur.wait_event(done)
        out["visited"] = p["visited"]
        return out

    fixed_stride = __import__("os").environ.get("CIS_FIXED_STRIDE", "1") != "0"   # device-side offsets + fixed-size payload all-gather (no host read per batch); False: the round-3 protocol
    stride_slack = 1.5    # x the even share nq * L / world (hits follow the query distribution, not the cell populations: +-20 % seen)

    def _exchange_and_merge(self, p, nq, L, check=True):
        """Exchange + merge of one partial result on the current stream.  check=True reads the overflow flag (one host read) and
        repeats the exchange with the exact size when a rank held more than the fixed stride; check=False leaves the flag in
        out["overflow"] for the caller to verify later (pipelined loops: one read after the loop, `overflowed(outs)`)."""
        import torch.distributed as dist
        from .lopq.search import merge_packed_dev
        on_device = p["packed"].is_cuda
        if self.fixed_stride and on_device and L > 0 and nq > 0:
            stride = exchange_stride(nq, L, self.world, self.stride_slack)
            parts, off, cnt_all, overflow = exchange_packed_fixed(p["packed"], p["cnt"], stride, self.group)
            out = merge_packed_dev(parts, off, cnt_all, nq, L)
            out["overflow"] = overflow
            if not check or int(overflow.item()) == 0:
                return out
        parts, off, cnt_all = exchange_packed(p["packed"], p["cnt"], self.group)  # exact size (one host read)
        return merge_packed_dev(parts, off, cnt_all, nq, L)  # HIP: one wave per query up to 3072, ranked places above

    @staticmethod
    def overflowed(outs):
        """True when any of the results of check=False calls was cut by the fixed exchange size (repeat those batches)."""
        import torch
        flags = [o["overflow"] for o in outs if "overflow" in o]
        return bool(flags) and bool(torch.stack([f.reshape(()) for f in flags]).max().item())

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
        out = self._exchange_and_merge(p, nq, L, check=True)
        out.pop("overflow", None)
        out["visited"] = p["visited"]
        return out


# ---- routed cell-sharded search (round 5) --------------------------------------------------------------------------------------------
# The all-gather protocol above hands every rank the whole batch: projection, cell ranking, the multisequence walk and the per-query
# part of the merge are done `world` times over, which is why one copy of the index over 8 GPUs projected to x1.9 only
# (profiles/archive/r04/r04f_shards.txt).  Routed:
#   home    every rank takes 1 / world of the batch, walks the multisequence against the cell sizes of the whole index and notes which
#           ranks own the non-empty cells each query visits (cis_index_query_owners_dev: one or two ranks at V = 16);
#   out     ONE all-to-all of fixed-size blocks carries each query to those owners (cis_route_queries_dev builds the blocks; the row
#           counts ride in a second, tiny all-to-all);
#   scan    an owner answers the queries it received with the ordinary partial search -- same walk, same quota cut, same visit ranks;
#   back    the ranked lists return to the home ranks (all-to-all, exact sizes) and are merged there by (dist, visit_rank, pos).
# A cell lives on one rank, so the merged list is the single index's, bit for bit.  Results stay with the home rank (rank r holds
# queries [nq r / world, nq (r + 1) / world)).  The host reads one small record per batch (rows sent / received) when the return trip
# is sized -- of an event two batches old in a pipelined loop.  A destination block that overflows sends the whole batch through the
# all-gather protocol instead (the flag is all-reduced: every rank takes the same branch).


def home_slice(nq, rank, world):
    """Queries [lo, hi) of a batch of nq that rank `rank` is the home of."""
    return (int(nq) * rank) // world, (int(nq) * (rank + 1)) // world


def route_capacity(nq_home, D, world, slack=2.0):
    """Rows of a destination block of the query all-to-all (D = 32-bit words per row).  Small rows: the whole home slice fits any
    block (no overflow possible).  Long rows (a 4096-d float32 query is 16 KB): `slack` x the even share of ~1.3 owners per query,
    overflow falls back."""
    nq_home = max(int(nq_home), 1)
    if world <= 1 or D <= 512:
        return nq_home
    return min(nq_home, int(slack * 1.3 * nq_home / world) + 64)


def route_slots_torch(mask, world, cap):
    """The tables of cis_route_queries_dev in torch (CPU tensors under gloo; the check of the HIP kernel): slot int32 [world, nq] --
    row of query i in the block for rank d, -1 = not sent --, cnt int32 [world], overflow int32 [1].  No host read."""
    import torch
    m = mask.to(torch.int64)
    bits = (m[None, :] >> torch.arange(world, dtype=torch.int64, device=m.device)[:, None]) & 1
    pos = torch.cumsum(bits, dim=1) - 1
    slot = torch.where((bits == 1) & (pos < cap), pos, torch.full_like(pos, -1)).to(torch.int32)
    tot = bits.sum(dim=1)
    return slot, torch.clamp(tot, max=cap).to(torch.int32), (tot > cap).any().to(torch.int32).reshape(1)


def route_rows_torch(q, slot, cap):
    """The send blocks [world, cap, D] for the tables of route_slots_torch (reference form; unused rows are zero)."""
    import torch
    world, nq = int(slot.shape[0]), int(slot.shape[1])
    out = torch.zeros((world, cap, q.shape[1]), dtype=q.dtype, device=q.device)
    for d in range(world):
        sel = slot[d] >= 0
        out[d, slot[d][sel].long()] = q[sel]
    return out


def _a2a(out, inp, out_splits=None, in_splits=None, group=None):
    """all_to_all_single; gloo has no all-to-all on device buffers: staged through the host there."""
    import torch.distributed as dist
    if dist.get_backend(group) != "nccl" and inp.is_cuda:
        o = out.cpu()
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    return out


def routed_send_queries(q_home, slot, send_q, cnt, overflow, group=None):
    """`out`: the fixed-size query all-to-all, the row counts, and the overflow flag agreed over the group.  Returns (recv_q
    [world, cap, D], recv_cnt int32 [world], overflow int32 [1]); tensors stay on the device of the inputs, no host read."""
    import torch
    import torch.distributed as dist
    recv_q = torch.empty_like(send_q)
    recv_cnt = torch.empty_like(cnt)
    _a2a(recv_cnt, cnt, group=group)
    _a2a(recv_q, send_q, group=group)
    ov = overflow.clone()
    if dist.get_backend(group) != "nccl" and ov.is_cuda:
        o = ov.cpu()
        dist.all_reduce(o, op=dist.ReduceOp.MAX, group=group)
        ov.copy_(o)
    else:
        dist.all_reduce(ov, op=dist.ReduceOp.MAX, group=group)
    return recv_q, recv_cnt, ov


def routed_return_hits(hits, n_recv, n_sent, group=None):
    """`back`: hits [sum(n_recv), L, 32] uint8 of the queries this rank answered (grouped by source rank, a source's rows in the
    order they arrived) go home.  n_recv / n_sent: rows received from / sent to every rank (host ints).  Returns [sum(n_sent), L, 32]:
    the lists of this rank's own queries, grouped by answering rank, in slot order."""
    import torch
    L = int(hits.shape[1])
    rec = L * 32
    out = torch.empty((int(sum(n_sent)), L, 32), dtype=torch.uint8, device=hits.device)
    _a2a(out.reshape(-1), hits.reshape(-1), out_splits=[int(c) * rec for c in n_sent], in_splits=[int(c) * rec for c in n_recv], group=group)
    return out


def routed_merge_tables(slot, n_sent, valid, L):
    """Where the list of home query i from rank d sits in the buffer routed_return_hits returned: off int64 [world, nq] (record
    offsets), cnt int32 [world, nq] (valid hits; 0 = rank d was not asked).  valid int32 [rows]: hits with id >= 0 per returned row."""
    import torch
    world, nq = int(slot.shape[0]), int(slot.shape[1])
    base = [0]
    for c in n_sent[:-1]:
        base.append(base[-1] + int(c))
    base_t = torch.tensor(base, dtype=torch.int64, device=slot.device)[:, None]
    row = torch.clamp(slot.to(torch.int64), min=0) + base_t
    sent = slot >= 0
    if valid.shape[0] == 0:
        cnt = torch.zeros((world, nq), dtype=torch.int32, device=slot.device)
    else:
        cnt = torch.where(sent, valid[torch.clamp(row, max=valid.shape[0] - 1)], torch.zeros_like(valid[:1])).to(torch.int32)
    off = torch.where(sent, row * L, torch.zeros_like(row))
    return off.contiguous(), cnt.contiguous()


def routed_merge_tables_dev(slot, n_sent, rec, L):
    """routed_merge_tables on the GPU in one launch (cis_routed_merge_tables_dev): rec int64 [rows * L, 4] = the returned lists."""
    import ctypes
    import torch
    from . import _lib
    world, nq = int(slot.shape[0]), int(slot.shape[1])
    base = (ctypes.c_int64 * world)()
    acc = 0
    for d in range(world):
        base[d] = acc
        acc += int(n_sent[d])
    off = torch.empty((world, nq), dtype=torch.int64, device=slot.device)
    cnt = torch.empty((world, nq), dtype=torch.int32, device=slot.device)
    _lib.check(_lib.lib().cis_routed_merge_tables_dev(slot.data_ptr(), world, nq, ctypes.cast(base, ctypes.c_void_p), rec.data_ptr(), int(L),
                                                      off.data_ptr(), cnt.data_ptr(), torch.cuda.current_stream(slot.device).cuda_stream))
    return off, cnt


class RoutedSearcher(object):
    """Cell-sharded search with every query routed to the owners of the cells it visits (see the comment above).

    Wraps a ShardedSearcher (its local index, lanes and process group).  ``search_begin(q_home)`` takes THIS rank's home slice of the
    batch (``home_slice``) and runs home + out asynchronously; ``search_end`` sizes the return trip (one host read), runs scan + back +
    merge and returns the results of the home slice, like ``search_batch_dev``.  At most ``sharded.pipeline_depth`` handles may be
    open at a time (a lane -- index view, stream, pinned record -- holds one batch)."""

    def __init__(self, sharded, slack=2.0):
        import torch.distributed as dist
        self.sh = sharded
        self.group = sharded.group
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self.slack = slack
        self.fallbacks = 0
        self._host = {}   # lane -> pinned record (rows sent / received, overflow) of the lane's batch in flight

    def search_begin(self, q_home, quota=10, limit=None, nq_total=None):
        """q_home: this rank's home slice (home_slice(nq_total, rank, world)) of a batch of nq_total queries.  nq_total sizes the blocks
        of the query all-to-all, which must be the same on every rank: pass it when the slices are ragged (default: world x this slice)."""
        import torch
        from . import _lib
        sv, stream = self.sh._lane()
        cur = torch.cuda.current_stream()
        st = stream if stream is not None else cur
        L = sv._dev_args(q_home, quota, limit)[0]
        nqh, D = int(q_home.shape[0]), int(q_home.shape[1])
        row_bytes = D * q_home.element_size()
        cap = route_capacity(-(-int(nq_total) // self.world) if nq_total is not None else nqh, row_bytes // 4, self.world, self.slack)
        dev = q_home.device
        if stream is not None:
            stream.wait_stream(cur)
            q_home.record_stream(stream)
        with torch.cuda.stream(st):
            mask, visited = sv.query_owners_dev(q_home, quota=quota)
            send_q = torch.empty((self.world, cap, D), dtype=q_home.dtype, device=dev)
            slot = torch.empty((self.world, nqh), dtype=torch.int32, device=dev)
            cnt = torch.empty(self.world, dtype=torch.int32, device=dev)
            overflow = torch.empty(1, dtype=torch.int32, device=dev)
            _lib.check(_lib.lib().cis_route_queries_dev(q_home.data_ptr(), nqh, row_bytes, mask.data_ptr(), self.world, cap, send_q.data_ptr(),
                                                        slot.data_ptr(), cnt.data_ptr(), overflow.data_ptr(), st.cuda_stream))
            recv_q, recv_cnt, ov = routed_send_queries(q_home, slot, send_q, cnt, overflow, self.group)
            host = self._host.get(id(sv))  # one page-locked record per lane (a lane holds one batch at a time: at most
            if host is None:               # `pipeline_depth` handles may be open, as for ShardedSearcher.search_begin)
                host = self._host[id(sv)] = torch.empty(2 * self.world + 1, dtype=torch.int32, pin_memory=True)
            host.copy_(torch.cat([cnt, recv_cnt, ov]), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        return {"sv": sv, "st": st, "q_home": q_home, "slot": slot, "recv_q": recv_q, "visited": visited, "host": host, "ev": ev,
                "quota": quota, "limit": limit, "L": L, "nqh": nqh}

    def search_end(self, h):
        import torch
        from .lopq.search import merge_packed_dev
        h["ev"].synchronize()  # the one host read of the batch: rows sent / received, overflow
        W = self.world
        host = h["host"].tolist()
        n_sent, n_recv, ov = host[:W], host[W:2 * W], host[2 * W]
        sv, st, L, nqh = h["sv"], h["st"], h["L"], h["nqh"]
        cur = torch.cuda.current_stream()
        if ov:  # a destination block overflowed somewhere: the whole batch through the all-gather protocol (same results)
            self.fallbacks += 1
            with torch.cuda.stream(st):
                q_all = all_gather_rows(h["q_home"], self.group)
                sizes = all_gather_stack(torch.tensor([nqh], dtype=torch.int64, device=q_all.device), self.group).reshape(-1).tolist()
                lo = int(sum(sizes[:self.rank]))
                p = sv.search_partial_packed_dev(q_all.contiguous(), quota=h["quota"], limit=h["limit"])  # this lane's handle and stream
                full = self.sh._exchange_and_merge(p, int(q_all.shape[0]), L, check=True)
                full.pop("overflow", None)
                out = {k: v[lo:lo + nqh] for k, v in full.items()}
                out["visited"] = p["visited"][lo:lo + nqh]
                done = torch.cuda.Event()
                done.record(st)
            cur.wait_event(done)
            return out
        with torch.cuda.stream(st):
            rows = torch.cat([h["recv_q"][s, :n_recv[s]] for s in range(W)]) if sum(n_recv) else h["recv_q"][0, :0]
            if rows.shape[0] and L > 0:
                hits, _ = sv.search_partial_dev(rows.contiguous(), quota=h["quota"], limit=h["limit"])
            else:
                hits = torch.empty((0, L, 32), dtype=torch.uint8, device=rows.device)
            back = routed_return_hits(hits, n_recv, n_sent, self.group)
            rec = back.reshape(-1).view(torch.int64).reshape(-1, 4)
            if rec.shape[0] == 0:
                rec = torch.zeros((1, 4), dtype=torch.int64, device=rows.device)
            off, cnt = routed_merge_tables_dev(h["slot"], n_sent, rec, L)
            out = merge_packed_dev(rec, off, cnt, nqh, L)
            out["visited"] = h["visited"]
            done = torch.cuda.Event(); done.record(st)
        for t in out.values():
            if hasattr(t, "record_stream"):
                t.record_stream(cur)
        cur.wait_event(done)
        return out

    def search_batch_dev(self, q_home, quota=10, limit=None, nq_total=None):
        return self.search_end(self.search_begin(q_home, quota=quota, limit=limit, nq_total=nq_total))


def all_gather_rows(x, group=None):
    """All-gather tensors that differ in their first dimension only -> the concatenation in group-rank order (same device).
    The row counts travel first (one small all-gather); the payload is padded to the largest count because neither RCCL nor
    gloo gathers ragged buffers.  gloo cannot gather device buffers: staged through the host there."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return x
    staged = dist.get_backend(group) != "nccl"
    dev = x.device
    work = x.cpu() if staged else x
    cnt = torch.tensor([x.shape[0]], dtype=torch.int64, device=work.device)
    cnts = all_gather_stack(cnt, group).reshape(-1).tolist()
    n_max = max(cnts)
    if x.shape[0] < n_max:
        pad = torch.zeros((n_max - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=work.device)
        work = torch.cat([work, pad])
    parts = all_gather_stack(work.contiguous(), group)  # [world, n_max, ...]
    if all(c == n_max for c in cnts):
        out = parts.reshape((world * n_max,) + tuple(x.shape[1:]))
    else:
        out = torch.cat([parts[r, :cnts[r]] for r in range(world)])
    return out.to(dev) if staged else out


def grid_groups(cell_shards):
    """Process groups of the 2-D layout: world = R query groups x S cell shards, world rank = g * S + s.
    Row group g = the S ranks that hold one whole copy of the index, sharded by cell (the all-gather merge runs inside
    it); column group s = the R ranks that hold the SAME cells in different copies.  Every rank must call this (the
    groups are created collectively, in the same order everywhere).  Returns (g, s, R, S, row_group, col_group); the
    groups are None where they are the whole world or a single rank."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    S = int(cell_shards)
    if S < 1 or world % S:
        raise ValueError("cell_shards=%d does not divide the world size %d" % (S, world))
    R = world // S
    g, s = rank // S, rank % S
    if S == 1 or R == 1:  # whole copies only (a row is one rank, the column the world), or one copy over the whole world
        return g, s, R, S, None, None
    rows = [dist.new_group(ranks=[gg * S + ss for ss in range(S)]) for gg in range(R)]
    cols = [dist.new_group(ranks=[gg * S + ss for gg in range(R)]) for ss in range(S)]
    return g, s, R, S, rows[g], cols[s]


class GridSearcher(object):
    """Queries sharded as well as cells: `world` ranks = R query groups x S cell shards.

    A query group is one ShardedSearcher (the index sharded by coarse cell over S ranks, partial search -> packed RCCL
    all-gather inside the group -> merge); the R groups hold identical copies and answer DIFFERENT queries, with no
    collective between them on the search path.  S is a capacity choice (the smallest S whose shard fits a GPU: a
    10M x M=8 index is 160 MB, so S=1 fits 288 GB thousands of times over), R = world / S is throughput: the front end
    (projection, cell ranking, tables), which cell sharding replicates on every rank, is divided by R.

    Build: every rank brings the slice it encoded; the slices are all-gathered along the column (the ranks that own the
    same cells in the other copies), then routed inside the row to the cell owners (ShardedSearcher.add_codes_routed_dev).
    Inside a cell the items arrive ordered by (row rank s of the source, column rank g, position in the slice): to get the
    order of a single index built from the whole data, give rank (g, s) the slice number s * R + g.
    """

    def __init__(self, model, cell_shards, owner=None):
        import torch.distributed as dist
        from .lopq.search import LOPQSearcherHIP
        self.g, self.s, self.R, self.S, row, col = grid_groups(cell_shards)
        self.world = dist.get_world_size()
        self._col = col if self.S > 1 else None     # S == 1: the column is the whole world (group None)
        if self.S > 1:
            self.row = ShardedSearcher(model, owner=owner, group=row)  # row None when R == 1: the world
            self.local = self.row.local
        else:
            self.row = None
            self.local = LOPQSearcherHIP(model)

    @property
    def slice_number(self):
        """Which of the `world` consecutive slices of the data this rank should encode (see the class comment)."""
        return self.s * self.R + self.g

    def query_group(self):
        return self.g, self.R

    def add_codes_dev(self, coarse, fine, ids, dedup=True):
        """This rank's encoded slice (tensors in HBM) into every copy of the index.  Returns what this rank accepted."""
        if self.R > 1:
            coarse = all_gather_rows(coarse, self._col)
            fine = all_gather_rows(fine, self._col)
            ids = all_gather_rows(ids, self._col)
        if self.row is not None:
            return self.row.add_codes_routed_dev(coarse, fine, ids, dedup=dedup)
        return self.local.add_codes_dev(coarse, fine, ids, dedup=dedup)[0]

    def get_nb_indexed(self):
        return self.local.get_nb_indexed()

    def search_begin(self, q, quota=10, limit=None):
        """q: the queries of THIS rank's query group (identical on the S ranks of the group)."""
        if self.row is not None:
            return self.row.search_begin(q, quota=quota, limit=limit)
        return {"out": self.local.search_batch_dev(q, quota=quota, limit=limit)}

    def search_end(self, h, check=True):
        if self.row is not None:
            return self.row.search_end(h, check=check)
        return h["out"]

    def search_batch_dev(self, q, quota=10, limit=None):
        if self.row is not None:
            return self.row.search_batch_dev(q, quota=quota, limit=limit)
        return self.local.search_batch_dev(q, quota=quota, limit=limit)
