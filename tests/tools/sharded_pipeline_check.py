"""One process per rank of an RCCL group (torchrun, or plain python = a world-1 group): the routed insert and the pipelined
sharded search (search_begin / search_end: exchange + merge of batch b on a side stream while batch b+1 is searched) equal
the plain single-GPU searcher.  With WORLD_SIZE > 1 every rank uses its own GPU (LOCAL_RANK) -- the form the N-GPU bench runs.
Usage: python tests/tools/sharded_pipeline_check.py   |   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 ... this file"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from conftest import load_golden
from test_lopq_hip_parity import hip_model
from columbiaimagesearch_amd import _lib
from columbiaimagesearch_amd.distributed import ShardedSearcher, greedy_cell_owner
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
backend = os.environ.get("CIS_CHECK_BACKEND", "nccl")  # "gloo": several ranks on ONE GPU (RCCL refuses shared devices): the staged collectives
if backend != "nccl":
    local = 0
torch.cuda.set_device(local)
_lib.check(_lib.lib().cis_set_device(local))
dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local) if backend == "nccl" else None)
z, X, Q = load_golden("c2")
m = hip_model(z)
coarse, fine = z["coarse"], z["fine"]
n = coarse.shape[0]
ids = np.arange(n, dtype=np.int64) + 7
# the single index sees the items in the order the sharded one does: two batches, each the concatenation of the ranks'
# slices (rank 0's first) -- per-cell insertion order decides between equal distances (duplicate codes are common here)
single = LOPQSearcherHIP(m)
bounds = [(r * n // world, (r + 1) * n // world) for r in range(world)]
for part in (0, 1):
    sel = np.concatenate([np.arange(a_, a_ + (b_ - a_) // 2 + 50) if part == 0 else np.arange(a_ + (b_ - a_) // 2, b_) for a_, b_ in bounds])
    single.add_codes_array(coarse[sel], fine[sel], ids[sel])
V = m.V
counts = np.bincount(coarse[:, 0].astype(np.int64) * V + coarse[:, 1], minlength=V * V)
sh = ShardedSearcher(m, owner=greedy_cell_owner(counts, world))
a, b = rank * n // world, (rank + 1) * n // world  # this rank's slice of the batch, as if it had encoded it
h1 = (b - a) // 2  # two batches, the second one repeating 50 items of the first (recognised as duplicates by their owners)
sh.add_codes_routed(coarse[a:a + h1 + 50], fine[a:a + h1 + 50], ids[a:a + h1 + 50])
sh.add_codes_routed(coarse[a + h1:b], fine[a + h1:b], ids[a + h1:b])
assert sh.get_nb_indexed() == single.get_nb_indexed(), (sh.get_nb_indexed(), single.get_nb_indexed())
# the same insert without a host copy: records packed on the device, one all-to-all of device buffers, device-side merge;
# the same two batches
sh_dev = ShardedSearcher(m, owner=greedy_cell_owner(counts, world))
cd, fd, idd = (torch.as_tensor(coarse[a:b].view(np.int16)).cuda(), torch.as_tensor(fine[a:b]).cuda(), torch.as_tensor(ids[a:b]).cuda())
sh_dev.add_codes_routed_dev(cd[:h1 + 50].contiguous(), fd[:h1 + 50].contiguous(), idd[:h1 + 50].contiguous())
sh_dev.add_codes_routed_dev(cd[h1:].contiguous(), fd[h1:].contiguous(), idd[h1:].contiguous())
assert sh_dev.get_nb_indexed() == single.get_nb_indexed(), (sh_dev.get_nb_indexed(), single.get_nb_indexed())
cc_a, cc_b = np.zeros(V * V, dtype=np.int64), np.zeros(V * V, dtype=np.int64)
_lib.check(_lib.lib().cis_index_cell_counts(sh.local._ix, _lib.ptr(cc_a)))
_lib.check(_lib.lib().cis_index_cell_counts(sh_dev.local._ix, _lib.ptr(cc_b)))
assert (cc_a == cc_b).all() and (cc_a == counts).all()
for c in range(0, V * V, 37):  # cells of this rank: same items in the same order whichever way they were inserted
    ga, gb = sh.local.get_cell((c // V, c % V)), sh_dev.local.get_cell((c // V, c % V))
    assert [i for i, _ in ga] == [i for i, _ in gb] and [x.fine for _, x in ga] == [x.fine for _, x in gb], c
if rank == 0:
    print("world %d: routed device insert == routed host insert" % world)
qs = [torch.as_tensor(Q[i:i + 16]).cuda().contiguous() for i in (0, 16, 32, 48)]
for quota, limit in [(3000, 100), (50, 20), (5000, 600), (20000, 3500)]:
    want = [single.search_batch_dev(q, quota=quota, limit=limit) for q in qs]
    got = []
    sh = sh_dev if limit == 20 else sh
    h = sh.search_begin(qs[0], quota=quota, limit=limit)
    for q in qs[1:]:
        h2 = sh.search_begin(q, quota=quota, limit=limit)
        got.append(sh.search_end(h))
        h = h2
    got.append(sh.search_end(h))
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        assert torch.equal(w["ids"], g["ids"]) and torch.equal(w["n_found"], g["n_found"]) and torch.equal(w["visited"], g["visited"])
        dw, dg = w["dists"], g["dists"]
        assert torch.equal(torch.isnan(dw), torch.isnan(dg)) and torch.equal(dw[~torch.isnan(dw)], dg[~torch.isnan(dg)])
    if rank == 0:
        print("world %d quota %d limit %d: routed insert + pipelined search ok" % (world, quota, limit))
# the exchange without a host read: fixed payload size + offsets by a kernel; the overflow flag is raised when a rank holds
# more than the fixed size (forced here with a tiny slack) and check=True then repeats the exchange with the exact size
for slack, expect_flag in [(1.5, None), (0.01, True)]:
    sh.stride_slack = slack
    want = [single.search_batch_dev(q, quota=3000, limit=100) for q in qs]
    outs = []
    for q in qs:
        outs.append(sh.search_end(sh.search_begin(q, quota=3000, limit=100), check=False))
    torch.cuda.synchronize()
    flagged = ShardedSearcher.overflowed(outs)
    if expect_flag is not None and world > 1:
        assert flagged == expect_flag, (slack, flagged)
    if not flagged:
        for w, g in zip(want, outs):
            assert torch.equal(w["ids"], g["ids"]) and torch.equal(w["n_found"], g["n_found"])
    for w, q in zip(want, qs):  # check=True: always exact, whatever the slack
        g = sh.search_end(sh.search_begin(q, quota=3000, limit=100))
        torch.cuda.synchronize()
        assert torch.equal(w["ids"], g["ids"]) and torch.equal(w["n_found"], g["n_found"]) and torch.equal(w["visited"], g["visited"])
        dw, dg = w["dists"], g["dists"]
        assert torch.equal(dw[~torch.isnan(dw)], dg[~torch.isnan(dg)])
sh.stride_slack = 1.5
if rank == 0:
    print("world %d: fixed-size exchange without a host read ok (overflow -> exact repeat)" % world)
# the routed search (round 5): every rank is the home of 1 / world of a batch, a query travels only to the owners of the cells it
# visits, results stay with the home rank -- equal to the single index on the home slice; pipelined over the lanes; and once more
# with destination blocks of one row (overflow -> the all-gather protocol answers the batch)
from columbiaimagesearch_amd.distributed import RoutedSearcher, home_slice, route_slots_torch, route_rows_torch
import columbiaimagesearch_amd.distributed as D_
rt = RoutedSearcher(sh_dev)
qall = torch.as_tensor(Q[:64]).float().cuda().contiguous()  # the routed rows travel as float32
# the HIP routing tables against their torch restatement
mask, vis = sh_dev.local.query_owners_dev(qall, quota=3000)
want_vis = single.search_batch_dev(qall, quota=3000, limit=10)["visited"]
assert torch.equal(vis, want_vis)
for cap in (64, 5):
    slot_t, cnt_t, ov_t = route_slots_torch(mask, world, cap)
    send = torch.full((world, cap, qall.shape[1]), -1.0, dtype=torch.float32, device="cuda")
    slot = torch.empty((world, 64), dtype=torch.int32, device="cuda"); cnt = torch.empty(world, dtype=torch.int32, device="cuda"); ov = torch.empty(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().cis_route_queries_dev(qall.data_ptr(), 64, qall.shape[1] * 4, mask.data_ptr(), world, cap, send.data_ptr(), slot.data_ptr(),
                                                cnt.data_ptr(), ov.data_ptr(), torch.cuda.current_stream().cuda_stream))
    assert torch.equal(slot, slot_t) and torch.equal(cnt, cnt_t) and torch.equal(ov, ov_t), cap
    rows_t = route_rows_torch(qall, slot_t, cap)
    used = (torch.arange(cap, device="cuda")[None, :] < cnt_t[:, None].long())
    assert torch.equal(send[used], rows_t[used])
assert int((mask != 0).sum()) == 64 and int(mask.max()) < (1 << world)
# the merge tables of the return trip: one launch against their torch restatement
from columbiaimagesearch_amd.distributed import routed_merge_tables, routed_merge_tables_dev
slot_t, cnt_t, _ = route_slots_torch(mask, world, 64)
n_sent_t = cnt_t.tolist()
fake = sh_dev.local.search_partial_dev(qall, quota=50, limit=20)[0]  # 64 ranked lists, some short
rows_back = fake[torch.arange(int(sum(n_sent_t)), device="cuda") % 64].contiguous()
rec_t = rows_back.reshape(-1).view(torch.int64).reshape(-1, 4)
valid_t = (rec_t[:, 2].reshape(-1, 20) >= 0).sum(dim=1, dtype=torch.int32)
o1, c1 = routed_merge_tables(slot_t, n_sent_t, valid_t, 20)
o2, c2 = routed_merge_tables_dev(slot_t, n_sent_t, rec_t, 20)
assert torch.equal(c1, c2) and torch.equal(o1[c1 > 0], o2[c1 > 0])
for quota, limit in [(3000, 100), (50, 20), (5000, 600), (20000, 3500), (10, 0)]:
    got, want = [], []
    batches = [qall[i:i + 16] for i in (0, 16, 32, 48)] + [qall[:7], qall]
    hs = []
    for qb in batches:
        lo, hi = home_slice(qb.shape[0], rank, world)
        w = single.search_batch_dev(qb.contiguous(), quota=quota, limit=limit)
        want.append({k: v[lo:hi] for k, v in w.items()})
        hs.append(rt.search_begin(qb[lo:hi].contiguous(), quota=quota, limit=limit, nq_total=qb.shape[0]))
        if len(hs) == 2:
            got.append(rt.search_end(hs.pop(0)))
    while hs:
        got.append(rt.search_end(hs.pop(0)))
    torch.cuda.synchronize()
    for bi, (w, g) in enumerate(zip(want, got)):
        for k in ("ids", "n_found", "visited"):
            assert torch.equal(w[k], g[k]), (rank, quota, limit, bi, k, (w[k] != g[k]).nonzero()[:4].tolist(), w[k].reshape(-1)[:6].tolist(), g[k].reshape(-1)[:6].tolist())
        dw, dg = w["dists"], g["dists"]
        assert torch.equal(torch.isnan(dw), torch.isnan(dg)) and torch.equal(dw[~torch.isnan(dw)], dg[~torch.isnan(dg)])
# float64 queries travel as float64 rows
q64 = torch.as_tensor(Q[:40]).double().cuda().contiguous()
lo, hi = home_slice(40, rank, world)
w = single.search_batch_dev(q64, quota=3000, limit=100)
g = rt.search_batch_dev(q64[lo:hi].contiguous(), quota=3000, limit=100, nq_total=40)
torch.cuda.synchronize()
assert torch.equal(w["ids"][lo:hi], g["ids"]) and torch.equal(w["visited"][lo:hi], g["visited"]) and torch.equal(w["dists"][lo:hi], g["dists"])
assert rt.fallbacks == 0
if rank == 0:
    print("world %d: routed search (owners only) == single index on every home slice" % world)
cap_fn = D_.route_capacity
D_.route_capacity = lambda nq_home, D, world, slack=2.0: 1
try:
    lo, hi = home_slice(64, rank, world)
    g = rt.search_batch_dev(qall[lo:hi].contiguous(), quota=3000, limit=100, nq_total=64)
    w = single.search_batch_dev(qall, quota=3000, limit=100)
    torch.cuda.synchronize()
    assert torch.equal(w["ids"][lo:hi], g["ids"]) and torch.equal(w["n_found"][lo:hi], g["n_found"]) and torch.equal(w["visited"][lo:hi], g["visited"])
    assert rt.fallbacks == 1
finally:
    D_.route_capacity = cap_fn
if rank == 0:
    print("world %d: routed search, overflowing block -> all-gather protocol ok" % world)
dist.barrier()
dist.destroy_process_group()
