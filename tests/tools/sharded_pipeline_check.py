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
torch.cuda.set_device(local)
_lib.check(_lib.lib().cis_set_device(local))
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
z, X, Q = load_golden("c2")
m = hip_model(z)
coarse, fine = z["coarse"], z["fine"]
n = coarse.shape[0]
ids = np.arange(n, dtype=np.int64) + 7
single = LOPQSearcherHIP(m); single.add_codes_array(coarse, fine, ids)
V = m.V
counts = np.bincount(coarse[:, 0].astype(np.int64) * V + coarse[:, 1], minlength=V * V)
sh = ShardedSearcher(m, owner=greedy_cell_owner(counts, world))
a, b = rank * n // world, (rank + 1) * n // world  # this rank's slice of the batch, as if it had encoded it
sh.add_codes_routed(coarse[a:b], fine[a:b], ids[a:b])
assert sh.get_nb_indexed() == single.get_nb_indexed(), (sh.get_nb_indexed(), single.get_nb_indexed())
qs = [torch.as_tensor(Q[i:i + 16]).cuda().contiguous() for i in (0, 16, 32, 48)]
for quota, limit in [(3000, 100), (50, 20), (5000, 600), (20000, 3500)]:
    want = [single.search_batch_dev(q, quota=quota, limit=limit) for q in qs]
    got = []
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
dist.barrier()
dist.destroy_process_group()
