"""One rank's share of a cell-sharded search on a single GPU (no collective): time of search_partial_dev and of the
packing, for world = 1, 2, 4, 8.  Usage: python tools/emulate_shard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = [sys.argv[0]]
import bench
from columbiaimagesearch_amd.distributed import greedy_cell_owner
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
from columbiaimagesearch_amd.lopq.search import merge_packed_dev
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ref_merge import pack_hits_dev
model, z = bench.load_model("c4")
dev = torch.device("cuda", 0)
P = bench.mixture_centers("descriptor", dev)
N = 10_000_000; chunk = N // 80
cs, fs = [], []
for c in range(80):
    co, fi = model.predict_batch_dev(bench.gen_chunk(P, c, chunk, dev)); cs.append(co); fs.append(fi)
coarse = torch.cat(cs).cpu().numpy().view(np.uint16); fine = torch.cat(fs).cpu().numpy()
cells = coarse[:, 0].astype(np.int64) * model.V + coarse[:, 1]
counts = np.bincount(cells, minlength=model.V * model.V)
q = bench.make_queries(bench.gen_chunk(P, 0, chunk, dev), 0, 8192, dev)
for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    owner = greedy_cell_owner(counts, world)
    s = LOPQSearcherHIP(model, shard=(0, world, owner))
    s.add_codes_array(coarse, fine, ids=np.arange(N, dtype=np.int64), dedup=False)
    for _ in range(2):
        h, v = s.search_partial_dev(q, quota=10000, limit=100)
    torch.cuda.synchronize()
    s.set_profiling(True)
    t = time.perf_counter(); K = 8
    for _ in range(K):
        h, v = s.search_partial_dev(q, quota=10000, limit=100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    prof = s.read_profile()
    t = time.perf_counter()
    for _ in range(K):
        pk, cnt = pack_hits_dev(h)
        off = (torch.cumsum(cnt[None], dim=1, dtype=torch.int64) - cnt[None]).contiguous()
        out = merge_packed_dev(pk[None].contiguous(), off, cnt[None].contiguous(), 8192, 100)
    torch.cuda.synchronize()
    dp = (time.perf_counter() - t) / K
    print("world %d rank 0: partial search %.3f ms (stages %s), pack+merge(1 list) %.3f ms, packed hits %d (%.1f MB)" % (
        world, dt * 1e3, {k: round(prof[k] / K, 3) for k in ("front_ms", "tables_ms", "scan_ms", "merge_ms")}, dp * 1e3,
        pk.shape[0], pk.shape[0] * 32 / 1e6))
    del s
