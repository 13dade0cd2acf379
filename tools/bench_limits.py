"""ms per 8192-query batch on the bench index for a sweep of `limit` (the routing thresholds of search_batch)."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench as B
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP

N = int(os.environ.get("CIS_BENCH_N", 10_000_000)); NCH = 10
dev = torch.device("cuda", 0)
model, z = B.load_model("c4")
P = B.mixture_centers("descriptor", dev)
co, fi = [], []
for c in range(NCH):
    a, b = model.predict_batch_dev(B.gen_chunk(P, c, N // NCH, dev)); co.append(a); fi.append(b)
s = LOPQSearcherHIP(model)
s.add_codes_array(torch.cat(co).cpu().numpy().view(np.uint16), torch.cat(fi).cpu().numpy(), ids=np.arange(N, dtype=np.int64), dedup=False)
x0 = B.gen_chunk(P, 0, N // NCH, dev)
nq = int(os.environ.get("NQ", 8192))
q = B.make_queries(x0, 0, nq, dev)
host = os.environ.get("HOST") == "1"  # host-pointer entry point (cis_index_search): PCIe copies in and out included
qh = q.cpu().numpy()
run = (lambda limit: s.search_batch(qh, quota=10000, limit=limit)) if host else (lambda limit: s.search_batch_dev(q, quota=10000, limit=limit))
for limit in [int(v) for v in os.environ.get("LIMITS", "10,100,440,441,952,953,3072,3073,10000").split(",")]:
    for _ in range(2):
        run(limit)
    torch.cuda.synchronize(); t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        run(limit)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print("limit %6d  %8.3f ms/batch  %10.0f queries/s" % (limit, ms, nq / ms * 1e3), flush=True)
