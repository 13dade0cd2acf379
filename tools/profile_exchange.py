"""Where the sharded step's exchange + merge time goes, on a world-1 RCCL group (one GPU): per-call host and device times of
search_begin (partial search), exchange_packed (counts all-gather, host read of the stride, packed all-gather) and the merge.
usage: python tools/profile_exchange.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
sys.argv = [sys.argv[0]]
import bench
from columbiaimagesearch_amd.distributed import ShardedSearcher, exchange_packed
from columbiaimagesearch_amd.lopq.search import merge_packed_dev
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model, z = bench.load_model("c4")
P = bench.mixture_centers("descriptor", dev)
N, chunks = 2_000_000, 16
sh = ShardedSearcher(model)
for c in range(chunks):
    x = bench.gen_chunk(P, c, N // chunks, dev)
    co, fi = model.predict_batch_dev(x)
    sh.add_codes_routed_dev(co, fi, torch.arange(c * (N // chunks), (c + 1) * (N // chunks), dtype=torch.int64, device=dev), dedup=False)
x0 = bench.gen_chunk(P, 0, N // chunks, dev)
q = bench.make_queries(x0, 0, 8192, dev)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out
t_part, p = timed(lambda: sh.local.search_partial_packed_dev(q, quota=10000, limit=100))
t_exch, ex = timed(lambda: exchange_packed(p["packed"], p["cnt"]))
parts, off, cnt_all = ex
t_merge, _ = timed(lambda: merge_packed_dev(parts, off, cnt_all, 8192, 100))
t_all, _ = timed(lambda: sh.search_batch_dev(q, quota=10000, limit=100))
def pipelined(reps=20):
    h = sh.search_begin(q, quota=10000, limit=100)
    for _ in range(reps - 1):
        h2 = sh.search_begin(q, quota=10000, limit=100)
        out = sh.search_end(h)
        h = h2
    return sh.search_end(h)
pipelined(3); torch.cuda.synchronize()
t = time.perf_counter(); pipelined(20); torch.cuda.synchronize(); t_pipe = (time.perf_counter() - t) / 20 * 1e3
t = time.perf_counter()
for _ in range(20):
    sh.search_begin(q, quota=10000, limit=100)
t_host_begin = (time.perf_counter() - t) / 20 * 1e3  # host time to enqueue a partial search (no synchronisation)
torch.cuda.synchronize()
print("pipelined begin/end: %.3f ms per step; host time of search_begin alone %.3f ms" % (t_pipe, t_host_begin))
print("world 1 (RCCL), 8192 queries, limit 100: partial search %.3f ms, exchange_packed %.3f ms, merge_packed %.3f ms, search_batch_dev %.3f ms"
      % (t_part, t_exch, t_merge, t_all))
dist.destroy_process_group()
