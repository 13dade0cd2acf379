"""One process per rank (torchrun): the 2-D layout of distributed.GridSearcher -- R query groups x S cell shards -- built from
per-rank slices (ragged: the slices differ in length) equals the plain single-GPU searcher, cell by cell (same items in the
same order) and answer by answer (every group its OWN queries, pipelined search_begin / search_end), for every S that
divides the world.  Backend: RCCL with one GPU per rank, or CIS_CHECK_BACKEND=gloo with all ranks on one GPU.
Usage: python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port P tests/tools/grid_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from conftest import load_golden
from test_lopq_hip_parity import hip_model
from columbiaimagesearch_amd import _lib
from columbiaimagesearch_amd.distributed import GridSearcher, greedy_cell_owner
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
backend = os.environ.get("CIS_CHECK_BACKEND", "nccl")
if backend != "nccl":
    local = 0
torch.cuda.set_device(local)
_lib.check(_lib.lib().cis_set_device(local))
dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local) if backend == "nccl" else None)
z, X, Q = load_golden("c2")
m = hip_model(z)
coarse, fine = z["coarse"], z["fine"]
n = coarse.shape[0]
ids = np.arange(n, dtype=np.int64) + 3
V = m.V
cell = coarse[:, 0].astype(np.int64) * V + coarse[:, 1]
counts = np.bincount(cell, minlength=V * V)
single = LOPQSearcherHIP(m)
single.add_codes_array(coarse, fine, ids, dedup=False)
# ragged slices: slice k = [cuts[k], cuts[k+1])
cuts = [0] + [int((k + 1) * n // world + (37 * (k + 1)) % 211 - 100) for k in range(world - 1)] + [n]
for S in [s_ for s_ in range(1, world + 1) if world % s_ == 0]:
    R = world // S
    gs = GridSearcher(m, S, owner=greedy_cell_owner(counts, S) if S > 1 else None)
    k = gs.slice_number
    assert (gs.g, gs.s) == (rank // S, rank % S) and k == (rank % S) * R + rank // S
    a, b = cuts[k], cuts[k + 1]
    cd, fd, idd = (torch.as_tensor(coarse[a:b].view(np.int16)).cuda(), torch.as_tensor(fine[a:b]).cuda(), torch.as_tensor(ids[a:b]).cuda())
    gs.add_codes_dev(cd, fd, idd, dedup=False)
    assert gs.get_nb_indexed() == n, (gs.get_nb_indexed(), n)
    cc = np.zeros(V * V, dtype=np.int64)
    _lib.check(_lib.lib().cis_index_cell_counts(gs.local._ix, _lib.ptr(cc)))
    assert (cc == counts).all()
    # the cells this rank holds: the single index's items in the single index's order
    owner = greedy_cell_owner(counts, S) if S > 1 else np.zeros(V * V, dtype=np.int32)
    mine = [c for c in range(V * V) if owner[c] == gs.s][::5]
    for c in mine:
        ga, gb = single.get_cell((c // V, c % V)), gs.local.get_cell((c // V, c % V))
        assert [i for i, _ in ga] == [i for i, _ in gb] and [x.fine for _, x in ga] == [x.fine for _, x in gb], (S, c)
    # every query group answers its own batches
    qs = [torch.as_tensor(Q[(4 * gs.g + i) * 8:(4 * gs.g + i + 1) * 8]).cuda().contiguous() for i in range(4)]
    for quota, limit in [(3000, 100), (50, 20), (20000, 3500)]:
        want = [single.search_batch_dev(q, quota=quota, limit=limit) for q in qs]
        got = []
        h = gs.search_begin(qs[0], quota=quota, limit=limit)
        for q in qs[1:]:
            h2 = gs.search_begin(q, quota=quota, limit=limit)
            got.append(gs.search_end(h))
            h = h2
        got.append(gs.search_end(h))
        got.append(gs.search_batch_dev(qs[0], quota=quota, limit=limit))
        want.append(want[0])
        torch.cuda.synchronize()
        for w, g_ in zip(want, got):
            assert torch.equal(w["ids"], g_["ids"]) and torch.equal(w["n_found"], g_["n_found"]) and torch.equal(w["visited"], g_["visited"])
            dw, dg = w["dists"], g_["dists"]
            assert torch.equal(torch.isnan(dw), torch.isnan(dg)) and torch.equal(dw[~torch.isnan(dw)], dg[~torch.isnan(dg)])
    dist.barrier()
    if rank == 0:
        print("world %d grid %d x %d: build and search equal the single index" % (world, R, S))
    del gs
dist.barrier()
dist.destroy_process_group()
