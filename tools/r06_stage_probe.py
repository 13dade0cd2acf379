import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np, torch, bench
from columbiaimagesearch_amd import _lib
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
device = torch.device("cuda", 0); torch.cuda.set_device(0); _lib.check(_lib.lib().cis_set_device(0))
model, z = bench.load_model("c4"); P = bench.mixture_centers("descriptor", device)
N = 200_000_000; n_chunks = 80; chunk_n = N // n_chunks
s = LOPQSearcherHIP(model); sub = 1 << 20
for c in range(n_chunks):
    x = bench.gen_chunk(P, c, chunk_n, device); co_l, fi_l = [], []
    for a in range(0, chunk_n, sub):
        co, fi = model.predict_batch_dev(x[a:a + sub]); co_l.append(co); fi_l.append(fi)
    ids = torch.arange(c * chunk_n, (c + 1) * chunk_n, dtype=torch.int64, device=device)
    s.add_codes_dev(torch.cat(co_l), torch.cat(fi_l), ids, dedup=False)
torch.cuda.synchronize()
x0 = bench.gen_chunk(P, 0, 1 << 20, device); q = bench.make_queries(x0, 0, 8, device)[:1].contiguous()
for quota in (N, 10000):
    for _ in range(3): s.search_batch_dev(q, quota=quota, limit=100)
    torch.cuda.synchronize()
    s.set_profiling(True); s.read_profile()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); s.search_batch_dev(q, quota=quota, limit=100); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    pr = s.read_profile(); s.set_profiling(False)
    print("quota", quota, "call median %.3f ms" % np.median(ts), {k: (round(v / 9, 4) if isinstance(v, float) else v) for k, v in pr.items()})
