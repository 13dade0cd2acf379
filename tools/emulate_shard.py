"""One rank's share of a cell-sharded search on a single GPU (no collective): time of search_partial_packed_dev and of the
merge of `world` lists, for world = 1, 2, 4, 8, and the projection for the R x S grids.  Usage: python tools/emulate_shard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = [sys.argv[0]]
import bench
from columbiaimagesearch_amd.distributed import greedy_cell_owner
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
from columbiaimagesearch_amd.lopq.search import merge_packed_dev
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
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
meas = {}
for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    owner = greedy_cell_owner(counts, world)
    s = LOPQSearcherHIP(model, shard=(0, world, owner))
    s.add_codes_array(coarse, fine, ids=np.arange(N, dtype=np.int64), dedup=False)
    for _ in range(2):
        p = s.search_partial_packed_dev(q, quota=10000, limit=100)
    torch.cuda.synchronize()
    s.set_profiling(True)
    t = time.perf_counter(); K = 8
    for _ in range(K):
        p = s.search_partial_packed_dev(q, quota=10000, limit=100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    prof = s.read_profile()
    # the merge this rank runs after the exchange: `world` lists of about its own list's size (its own list, `world` times:
    # the right number of lists and the right volume, nq x L hits in total; the answer is meaningless, the time is not)
    total = int(p["total"].item())
    parts = p["packed"][:total][None].expand(world, total, 4).contiguous()
    off = p["off"][None].expand(world, 8192).contiguous()
    cnt = p["cnt"][None].expand(world, 8192).contiguous()
    merge_packed_dev(parts, off, cnt, 8192, 100)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        out = merge_packed_dev(parts, off, cnt, 8192, 100)
    torch.cuda.synchronize()
    dp = (time.perf_counter() - t) / K
    # realistic merge input: this rank's list in ONE slot and empty lists in the others for the queries whose hits are all here, i.e.
    # what a V = 16 index produces (a query's one or two cells live on one or two ranks); lists = [mine, empty, ...]
    cnt1 = torch.zeros_like(cnt); cnt1[0] = p["cnt"]
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K):
        out = merge_packed_dev(parts, off, cnt1, 8192, 100)
    torch.cuda.synchronize()
    dp1 = (time.perf_counter() - t) / K
    # three batches in flight (views on three streams), partial search + merge per batch: what a rank sustains per step
    lanes = [(s, torch.cuda.current_stream())] + [(s.view(), torch.cuda.Stream()) for _ in range(2)]
    def one(i):
        sv, stream = lanes[i % 3]
        with torch.cuda.stream(stream):
            pp = sv.search_partial_packed_dev(q, quota=10000, limit=100)
            merge_packed_dev(parts, off, cnt1, 8192, 100)
    for i in range(6): one(i)
    torch.cuda.synchronize(); t = time.perf_counter(); K2 = 24
    for i in range(K2): one(i)
    torch.cuda.synchronize()
    dpipe = (time.perf_counter() - t) / K2
    print("world %d rank 0: partial search, packed %.3f ms (stages %s), merge of %d full lists %.3f ms / of one list + %d empty %.3f ms, packed hits of this rank %d (%.1f MB); "
          "pipelined (3 in flight) partial search + merge %.3f ms per step" % (
        world, dt * 1e3, {k: round(prof[k] / K, 3) for k in ("front_ms", "tables_ms", "scan_ms", "merge_ms")}, world, dp * 1e3, world - 1, dp1 * 1e3,
        total, total * 32 / 1e6, dpipe * 1e3))
    meas[world] = (dt * 1e3, dp1 * 1e3 if world > 1 else 0.0, dpipe * 1e3)
    for sv, _ in lanes[1:]: sv.close()
    del s
# projection for the R x S grid of distributed.GridSearcher: a GPU spends partial(S) + pack/merge(S) per batch of 8192 queries
# of ITS query group (the exchange runs on the side stream under the next batch's search); R groups work side by side
print("projected whole-job time per 8192 queries (ms) and speed-up over one GPU:")
base = meas[1][0]
for n in (1, 2, 4, 8):
    row = []
    for S in sorted(set([1, min(2, n), n])):
        if S in meas and n % S == 0:
            t = (meas[S][0] + meas[S][1]) / (n // S)
            tp = meas[S][2] / (n // S)
            row.append("%d groups x %d shards: %.3f (x%.2f), pipelined %.3f (x%.2f)" % (n // S, S, t, base / t, tp, meas[1][2] / tp))
    print("  %d GPU(s): %s" % (n, "; ".join(row)))
