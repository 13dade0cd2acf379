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
    if world > 1:
        # The ROUTED protocol's share of rank 0 (distributed.RoutedSearcher; the two all-to-alls are not emulated): home work for its
        # 1 / world of the batch (owners of the visited cells + the send blocks), the partial search of the queries that visit rank
        # 0's cells -- taken from the WHOLE batch: what the other ranks would send --, the merge of its home queries' lists.
        # Twice: the bench's batch of 8192 queries over the whole job, and 8192 queries PER RANK (the regime routing is for: the
        # all-gather protocol's every rank projects, ranks and walks all world x 8192 queries).
        from columbiaimagesearch_amd import _lib
        from columbiaimagesearch_amd.distributed import home_slice, route_capacity, routed_merge_tables_dev
        def timed(f, K=8):
            f(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K): f()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
        lanes = [(s, torch.cuda.current_stream())] + [(s.view(), torch.cuda.Stream()) for _ in range(2)]
        for per_rank in ([False, True] if world >= 4 else [False]):
            qg = torch.cat([bench.make_queries(bench.gen_chunk(P, 0, chunk, dev), b, 8192, dev) for b in range(world)]) if per_rank else q
            nq = int(qg.shape[0])
            lo, hi = home_slice(nq, 0, world)
            qh = qg[lo:hi].contiguous()
            mask_all, _ = s.query_owners_dev(qg, quota=10000)
            rows0 = qg[(mask_all & 1) != 0].contiguous()
            owners_per_query = float(sum(int(((mask_all >> r) & 1).sum()) for r in range(world))) / nq
            cap = route_capacity(hi - lo, qg.shape[1] * qg.element_size() // 4, world)
            send_q = torch.empty((world, cap, qg.shape[1]), dtype=qg.dtype, device=dev)
            slot = torch.empty((world, hi - lo), dtype=torch.int32, device=dev); rcnt = torch.empty(world, dtype=torch.int32, device=dev); rov = torch.empty(1, dtype=torch.int32, device=dev)
            def home(sv):
                mk, vis = sv.query_owners_dev(qh, quota=10000)
                _lib.check(_lib.lib().cis_route_queries_dev(qh.data_ptr(), hi - lo, qg.shape[1] * qg.element_size(), mk.data_ptr(), world, cap, send_q.data_ptr(), slot.data_ptr(),
                                                            rcnt.data_ptr(), rov.data_ptr(), torch.cuda.current_stream().cuda_stream))
            def scan(sv):
                return sv.search_partial_dev(rows0, quota=10000, limit=100)[0]
            home(s); hits = scan(s); torch.cuda.synchronize()
            n_sent = rcnt.tolist()
            need = int(sum(n_sent))
            back = hits[torch.arange(need, device=dev) % hits.shape[0]].contiguous()  # as many lists as this rank's queries were sent out
            rec = back.reshape(-1).view(torch.int64).reshape(-1, 4)
            def merge():
                off_r, cnt_r = routed_merge_tables_dev(slot, n_sent, rec, 100)
                return merge_packed_dev(rec, off_r, cnt_r, hi - lo, 100)
            t_home, t_scan, t_merge = timed(lambda: home(s)), timed(lambda: scan(s)), timed(merge)
            def one_r(i):
                sv, stream = lanes[i % 3]
                with torch.cuda.stream(stream):
                    home(sv); scan(sv); merge()
            for i in range(6): one_r(i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(24): one_r(i)
            torch.cuda.synchronize()
            t_rpipe = (time.perf_counter() - t0) / 24 * 1e3
            # the all-gather protocol's share of rank 0 for the same global batch
            t_ag = timed(lambda: s.search_partial_packed_dev(qg, quota=10000, limit=100)) if per_rank else dt * 1e3
            def one_a(i):
                sv, stream = lanes[i % 3]
                with torch.cuda.stream(stream):
                    sv.search_partial_packed_dev(qg, quota=10000, limit=100)
            t_agpipe = dpipe * 1e3
            if per_rank:
                for i in range(6): one_a(i)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(12): one_a(i)
                torch.cuda.synchronize()
                t_agpipe = (time.perf_counter() - t0) / 12 * 1e3
            print("world %d rank 0 ROUTED, %d queries per step over the job: %.2f owners per query, %d queries reach this rank; home (owners + send blocks, %d queries) %.3f ms, partial search of the "
                  "received queries %.3f ms, merge of the home queries %.3f ms: %.3f ms serial, pipelined (3 in flight) %.3f ms per step | all-gather protocol, same batch: partial search %.3f ms serial, "
                  "pipelined %.3f ms (+ its merge)" % (world, nq, owners_per_query, rows0.shape[0], hi - lo, t_home, t_scan, t_merge, t_home + t_scan + t_merge, t_rpipe, t_ag, t_agpipe))
            if not per_rank:
                meas[world] = meas[world] + (t_home + t_scan + t_merge, t_rpipe)
            else:
                one = meas[1][2] * world  # one GPU answers world x 8192 queries in world steps
                print("  -> %d queries per step: one GPU %.3f ms (pipelined), %d GPUs all-gather %.3f ms (x%.2f), routed %.3f ms (x%.2f)" % (
                    nq, one, world, t_agpipe, one / t_agpipe, t_rpipe, one / t_rpipe))
        for sv, _ in lanes[1:]: sv.close()
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
            txt = "%d groups x %d shards: %.3f (x%.2f), pipelined %.3f (x%.2f)" % (n // S, S, t, base / t, tp, meas[1][2] / tp)
            if len(meas[S]) > 3:
                txt += ", ROUTED %.3f (x%.2f), pipelined %.3f (x%.2f)" % (meas[S][3] / (n // S), base / (meas[S][3] / (n // S)), meas[S][4] / (n // S), meas[1][2] / (meas[S][4] / (n // S)))
            row.append(txt)
    print("  %d GPU(s): %s" % (n, "; ".join(row)))
